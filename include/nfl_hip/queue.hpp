// nfl_hip/queue.hpp -- part of the drop-in header; include <nfl_hip/nfl.hpp> (or the reference's names under include/nfl*).
// payloads of resident handles and the deferred queue: recording, dependency levelling, grouping, transform fusion.
#ifndef NFL_HIP_QUEUE_HPP
#define NFL_HIP_QUEUE_HPP
#ifndef NFL_HIP_NFL_HPP
#error "include <nfl_hip/nfl.hpp>: the parts depend on each other in its order"
#endif
namespace nfl {
template <class T, size_t Degree, size_t NbModuli> class poly;
template <class T, size_t Degree, size_t NbModuli> class poly_p;
namespace detail {
inline std::atomic<bool> &deferred_flag();
}
inline void set_deferred(bool on) { detail::deferred_flag().store(on); }
namespace tests {
template <class P> class poly_tests_proxy;  // (poly.hpp:69-76) defined by the caller's test code, befriended below
}

namespace detail {
#ifdef NFL_HIP_REFERENCE_WORDS
static constexpr int dist_flags = NFLHIP_DIST_REFERENCE_WORDS;
#else
static constexpr int dist_flags = 0;
#endif

template <class P> struct lazy;

// Handle payloads come and go at the rate of the caller's temporaries (three per encryption of the LWE demo loop), and a
// general-purpose malloc / free pair per payload was the largest single item of the per-polynomial host cost (tools/hostprof:
// ~115 ns of ~185 per temporary).  std::allocate_shared with this allocator takes the control block + payload from a
// per-thread free list of fixed-size blocks instead; blocks freed on another thread simply join that thread's list.
template <class U> struct block_pool_alloc {
  typedef U value_type;
  block_pool_alloc() noexcept {}
  template <class V> block_pool_alloc(const block_pool_alloc<V> &) noexcept {}
  template <class V> struct rebind { typedef block_pool_alloc<V> other; };
  struct node { node *next; };
  struct list_t {
    node *head;
    size_t count;
    list_t() : head(nullptr), count(0) {}
    ~list_t() {
      gone() = true;   // (payloads released later in this thread's teardown -- static destructors -- go straight back to the heap)
      while (head) {
        node *n = head;
        head = n->next;
        ::operator delete(static_cast<void *>(n));
      }
      count = 0;
    }
  };
  static bool &gone() {   // trivially destructible, so it outlives the list it guards
    static thread_local bool g = false;
    return g;
  }
  static list_t &list() {
    static thread_local list_t l;
    return l;
  }
  U *allocate(size_t n) {
    if (n == 1 && sizeof(U) >= sizeof(node) && !gone()) {
      list_t &l = list();
      if (l.head) {
        node *x = l.head;
        l.head = x->next;
        --l.count;
        return reinterpret_cast<U *>(x);
      }
    }
    return static_cast<U *>(::operator new(n * sizeof(U)));
  }
  void deallocate(U *p, size_t n) noexcept {
    if (n == 1 && sizeof(U) >= sizeof(node) && !gone()) {
      list_t &l = list();
      if (l.count < (size_t(1) << 16)) {   // (bounded: a burst of 65 536 dead temporaries is kept, the rest goes back)
        node *x = reinterpret_cast<node *>(p);
        x->next = l.head;
        l.head = x;
        ++l.count;
        return;
      }
    }
    ::operator delete(static_cast<void *>(p));
  }
  template <class V> bool operator==(const block_pool_alloc<V> &) const noexcept { return true; }
  template <class V> bool operator!=(const block_pool_alloc<V> &) const noexcept { return false; }
};

// The shared payload of a poly_p handle (poly_p.hpp:11-204 keeps a std::shared_ptr<poly>): one polynomial that lives
// in HBM (`dev`), on the host (`host`), or both.  host_valid / dev_valid say which image holds the current value;
// neither valid = the zero polynomial (what poly_p() is) with nothing allocated yet.  Every device-side operation is
// enqueued on the context's stream, so the only synchronisation points are the device-to-host copies below.
// queued(): the value is the result of operations that are still in the deferred queue (lazy<P> below: recorded, or handed to
// a queue run that has not been retired yet); every access other than enqueueing more work runs the queue first (pending()).
// A queue run (lazy<P>::execute, possibly on the queue's own thread) never touches a payload: take() copies what it needs into
// per-run arrays and retire() writes the buffers it assigned back.  Every field belongs to the recording threads (under the
// queue's lock), which in turn leave `dev` of a queued value alone until its run is retired.
template <class P> struct payload : std::enable_shared_from_this<payload<P>> {
  typedef typename P::value_type T;
  typedef context<T, P::degree, P::nmoduli> ctx_t;
  static constexpr size_t bytes = sizeof(T) * P::degree * P::nmoduli;
  P *host;
  void *dev;
  bool host_valid, dev_valid;
  unsigned qrun;  // the recording run of the last deferred operation that writes this value (0: none); see queued()
  bool poisoned;  // the deferred operation that was to produce this value never ran (an earlier launch of its queue run failed)
  long qrefs;  // references the deferred queue holds to this value (one per queue run that mentions it: the one being recorded, the
              // one in flight): copy-on-write decisions look past them
  unsigned pin_at;  // where the recording run's reference to this payload sits in its pin list (valid while rec_run is the current one)
  // recording scratch of lazy<P>::record (valid when `rec_run` is the queue's current recording run): index of the last
  // recorded operation that writes / reads this value -- what lets a transform join the operation that produced its operand
  unsigned rec_run;
  int rec_w, rec_r;

  payload() : host(nullptr), dev(nullptr), host_valid(false), dev_valid(false), qrun(0), poisoned(false), qrefs(0), pin_at(0), rec_run(0), rec_w(-1), rec_r(-1) {}
  payload(const payload &o) : std::enable_shared_from_this<payload<P>>(), host(nullptr), dev(nullptr), host_valid(false),
                              dev_valid(false), qrun(0), poisoned(false), qrefs(0), pin_at(0), rec_run(0), rec_w(-1), rec_r(-1) {
    pending();
    o.usable();
    if (o.dev_valid) {  // stays on the device
      check(ctx(), nflhip_memcpy_d2d(ctx(), dev_wo(), o.dev, bytes, ctx_t::queue()), "poly_p copy");
    } else if (o.host_valid) {
      alloc_host();
      std::memcpy(host->data(), o.host->cdata(), bytes);
      host_valid = true;
    }
  }
  payload &operator=(const payload &) = delete;
  ~payload() {
    if (host) {
      host->~P();
      free(host);
    }
    ctx_t::release(dev);
  }
  static nflhip_ctx *ctx() { return ctx_t::get(); }
  static void pending() { lazy<P>::inst().flush(); }  // run whatever is still deferred
  bool queued() const { return qrun != 0 && qrun > lazy<P>::inst().done_run_; }  // (read under the queue's lock, or after pending())
  void usable() const {  // reading a value whose producing operation never ran is an error, not stale HBM
    if (poisoned) throw std::runtime_error("nfl(hip): this polynomial's deferred operation did not run (an earlier operation of its queue failed)");
  }

  void alloc_host() {
    if (host) return;
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(P)) != 0) throw std::bad_alloc();
    host = new (mem) P(uninitialized_t());
  }
  // the host image, current
  void to_host() {
    pending();
    usable();
    alloc_host();
    if (host_valid) return;
    if (dev_valid) {
      check(ctx(), nflhip_memcpy_d2h(ctx(), host->data(), dev, bytes, ctx_t::queue()), "poly_p download");
      check(ctx(), nflhip_stream_sync(ctx(), ctx_t::queue()), "poly_p download");
    } else {
      std::memset(static_cast<void *>(host->data()), 0, bytes);
    }
    host_valid = true;
  }
  P &host_rw() {  // the caller may write through the reference: the device image goes stale
    to_host();
    dev_valid = false;
    return *host;
  }
  P const &host_ro() {
    to_host();
    return *host;
  }
  P &host_wo() {  // about to be overwritten entirely on the host
    pending();
    alloc_host();
    host_valid = true;
    dev_valid = false;
    poisoned = false;
    return *host;
  }
  // the device image, current
  const void *dev_ro() {
    pending();
    return dev_ro_nf();
  }
  const void *dev_ro_nf() {  // (the queue's own form: never runs the queue)
    if (queued()) return nullptr;  // produced by a deferred operation; its buffer is assigned when the queue runs (and is the run's until then)
    usable();
    if (!dev) dev = ctx_t::acquire();
    if (!dev_valid) {
      if (host_valid) check(ctx(), nflhip_memcpy_h2d(ctx(), dev, host->cdata(), bytes, ctx_t::queue()), "poly_p upload");
      else check(ctx(), nflhip_memset_dev(ctx(), dev, 0, bytes, ctx_t::queue()), "poly_p zero");
      dev_valid = true;
    }
    return dev;
  }
  void *dev_rw() {  // in-place device operation
    dev_ro();
    host_valid = false;
    return dev;
  }
  void *dev_wo() {  // about to be overwritten entirely on the device
    pending();
    if (!dev) dev = ctx_t::acquire();
    dev_valid = true;
    host_valid = false;
    poisoned = false;
    return dev;
  }
};

// ---------------------------------------------------------------- deferred execution of per-polynomial operations
// One polynomial is 4 workgroups of work: a kernel launched for it runs for ~15 us on an otherwise empty GPU, and code
// written against the reference issues exactly such operations in loops (tests/nfllib_demo_main_op.cpp:26-58: three
// Gaussian polynomials, three transforms and two fused multiply-adds per encryption, one polynomial at a time).  So
// operations on resident handles are not launched when they are called: they are appended to a per-ring queue, and
// when a value is needed on the host (or the queue is long) the queue runs as a few BATCHED launches:
//   * operations are levelled by their data dependencies (read-after-write, write-after-read, write-after-write on
//     the shared payloads), so everything inside a level is independent;
//   * inside a level, operations with the same signature (same expression program, same distribution and parameters,
//     same transform) form a group; operands that are the same polynomial throughout (a key) split groups;
//   * results that have no buffer yet get CONSECUTIVE buffers (context::acquire_many), so a loop's temporaries form
//     dense arrays; a group is cut into runs whose operands advance by constant strides and every run is ONE launch of
//     the batch entry points (nflhip_eval_strided_dev, nflhip_sample[_gauss]_seq_dev, nflhip_ntt_fwd/inv_dev).
// Results are bit-identical to immediate execution (the random constructors keep the stream id they took when they
// were called).  nfl::set_deferred(false) (or -DNFL_HIP_EAGER) launches every operation at once instead.
inline std::atomic<bool> &deferred_flag() {
#ifdef NFL_HIP_EAGER
  static std::atomic<bool> f(false);
#else
  static std::atomic<bool> f(true);
#endif
  return f;
}

template <class P> struct lazy {
  typedef payload<P> pay_t;
  typedef typename pay_t::ctx_t ctx_t;
  typedef typename P::value_type T;
  typedef std::shared_ptr<pay_t> ptr_t;
  // K_FWD_FMA / K_FMA_INV / K_NOP are never recorded: a queue run rewrites recorded sequences into them (fuse())
  enum kind_t { K_EVAL = 0, K_NTT_FWD, K_NTT_INV, K_SAMPLE, K_GAUSS, K_FILL, K_FWD_FMA, K_FMA_INV, K_NOP };
  // One recorded operation: 120 bytes, trivially destructible.  The payloads it names are kept alive by ONE reference per
  // payload and queue run (`pins`), not one per mention -- a loop's temporaries are mentioned three times each.
  static constexpr int max_in = 4;  // expressions with more distinct handle operands are launched at once, not recorded
  struct op {
    pay_t *out;
    union {
      struct {
        pay_t *in[max_in];
        unsigned char code[NFLHIP_EXPR_MAX_LEN];
      } e;          // K_EVAL (the transforms use out only)
      struct {
        uint64_t p0, p1, sid;
        const nflhip_gauss *tab;
        int dist;
      } s;          // K_SAMPLE, K_GAUSS, K_FILL
      struct {
        pay_t *in[max_in];     // the key operands k0 [, k1] (same place as e.in: the levelling reads them through it)
        pay_t *out2;           // second result (out1 = NTT(x) * k1 + NTT(e1)), or nullptr
        uint64_t sid[3];       // stream ids of the Gaussian polynomials x, e0, e1
        uint32_t amp[3];       // their amplifiers
        unsigned out2_pin;
        const nflhip_gauss *tab;
      } f;          // K_FWD_FMA
    };
    unsigned char kind, nin, len;
    unsigned char post;   // 0, or K_NTT_FWD / K_NTT_INV: the result is transformed in place right after (a transform recorded on
                          // a value nothing had read since this operation produced it joins the operation instead of becoming a record)
    // Where the run's references to `out` and to the inputs sit in its pin list.  A queue run works on THESE: whatever it keeps
    // per value (buffer, levels, last writer) lives in arrays of its own, indexed by pin -- it never touches a payload, whose
    // cache lines stay with the recording thread (the first version of the queue's thread wrote its levelling scratch into the
    // payloads: the recording thread then fetched every line back from the other core when it retired the run, 1.5 us per LWE
    // encryption, 2.5 times slower than no thread at all).
    unsigned out_pin, in_pin[max_in];
  };
  // ---- the recording side (under mu).  The three groups below sit on cache lines of their own: the queue's thread polls `st` and
  // writes its statistics while the recording thread appends to q with every operation -- on one line that costs the recorder a
  // cross-core transfer per record (measured: the LWE loop recorded 2.5 times SLOWER with the groups interleaved).
  alignas(64) light_lock mu;
  std::vector<op> q;           // recorded operations
  std::vector<ptr_t> pins;     // the payloads they name, one reference each
  unsigned rec_run_;           // the recording run: bumped whenever the queue is handed to a queue run (payload::rec_run)
  std::atomic<unsigned> done_run_;  // the last recording run that has been retired (payload::queued)
  size_t next_run_;            // records at which the next run is handed to the queue's thread
  std::thread *th;             // the queue's thread, created by the first hand-over (leaked in a forked child, which has no such thread)
  long th_pid;
  bool th_failed;
  // ---- the handshake
  alignas(64) std::atomic<int> st;   // 0: no run in flight; 1: `fly` is with the queue's thread; 2: it is done with it (retire() is due)
  std::atomic<bool> w_sleeping, u_sleeping, quit;
  alignas(64) std::mutex wm;
  std::condition_variable wcv, ucv;
  // ---- the executing side
  // The queue run in flight: the records and pins of one recording run (their vectors swap with q / pins: no regrowth), from
  // take() to retire().  Between post() and the moment `st` turns 2 it belongs to the queue's thread; otherwise to whoever
  // holds `mu`.
  struct alignas(64) run_t {
    std::vector<op> ops, work;
    bool remote;               // executed by the queue's thread
    std::vector<ptr_t> held;
    std::vector<unsigned char> launched;
    // per pin, filled by take() (recording thread; nothing is in flight then, so every payload's `dev` is final): the value's
    // buffer (nullptr: none yet -- the run assigns one to its results and retire() writes it back) and whether the run's pin is
    // the LAST reference (no handle left: what the transform fusion needs to know before it lets a temporary never exist)
    std::vector<void *> dev;
    std::vector<unsigned char> dead;
    std::vector<int> wlev, rlev, fw;   // the run's levelling scratch, per pin
    unsigned char key[32];     // the sampler key as it was when the run was taken: set_sampler_key runs the queues before it changes it
    unsigned id;               // its recording run (payload::qrun)
    bool complete;
    std::exception_ptr error;  // what stopped it (rethrown by retire())
    run_t() : remote(false), id(0), complete(false) {}
  } fly;
  std::atomic<size_t> launches, coalesced;  // statistics: launches issued / operations they carried
  std::atomic<size_t> fused_fwd, fused_inv;  // statistics: sequences rewritten so far
  // compact Gaussian polynomials of a fused run live in ONE grow-only device buffer (every consumer is on the queue's stream)
  void *small_;
  size_t small_cap_;
  alignas(64) char pad_[64];   // (the executing side ends on a line of its own: whatever follows this object is not ours to slow down)
  // records after which the queue runs by itself: long enough for wide launches, short enough that the device works on
  // one part of a loop while the host records the next (NFL_HIP_QUEUE_LIMIT overrides, for experiments).  Measured on the
  // LWE demo loop (profiles/r02_late_queue_limit.txt): 2 048 iterations 614 k / 738 k / 1.04 M / 861 k encryptions/s with
  // 2 048 / 4 096 / 8 192 / 16 384 records, 16 384 iterations 1.09 M / 1.21 M / 1.16 M / 1.16 M with 4 096 ... 32 768.
  static size_t max_queue() {
    static const size_t v = getenv("NFL_HIP_QUEUE_LIMIT") ? size_t(atol(getenv("NFL_HIP_QUEUE_LIMIT"))) : 8192;
    return v ? v : 1;
  }
  // ---- the queue's own thread (round 6).  Preparing a queue run -- fusing, levelling, grouping, finding the stride runs -- and
  // issuing its launches cost the recording thread as much as the recording itself (tools/hostprof: 0.2 us of 0.43 us per LWE
  // encryption).  A run that starts because the queue is long enough is therefore handed to a thread of the queue's own:
  // the recording thread swaps in the other pair of vectors and goes on recording while that thread works.  ONE run is in flight
  // at most; the recording thread retires it (drops the pins, rethrows what stopped it) when it hands over the next one or when
  // anything needs the queue empty.  A run that somebody WAITS for (flush(): a value is read on the host, synchronize()) is
  // executed by the waiting thread itself once the one in flight is retired: no hand-over latency on short programs.
  // With a thread to hand runs to, the queue does not wait for max_queue() records either: from min_run() records on it hands
  // over as soon as the thread is idle, so the device starts on a loop's first part while the host is still recording
  // (the 2 048-iteration LWE loop used to leave the device idle for four fifths of its recording).
  // NFL_HIP_QUEUE_THREAD=0 (or a machine with one hardware thread, or a thread that cannot be created): every run is executed by
  // the thread that starts it, as before.  Results are identical either way (tests/cpp: deferred == immediate under both).
  static bool threaded() {
    static const bool v = getenv("NFL_HIP_QUEUE_THREAD") ? atoi(getenv("NFL_HIP_QUEUE_THREAD")) != 0 : std::thread::hardware_concurrency() > 1;
    return v;
  }
  static size_t min_run() {
    static const size_t v = getenv("NFL_HIP_QUEUE_MIN") ? size_t(atol(getenv("NFL_HIP_QUEUE_MIN"))) : 1024;
    return v ? v : 1;
  }
  // (Round 5 measured two run-length policies for SHORT loops and dropped both -- LWE demo loop, encryptions/s against the fixed
  //  length: a loop's first two runs at HALF length: 2 048 iterations 1.68 -> 1.85 M, but 1 024: 1.60 -> 1.39 M, 4 096: 2.12 ->
  //  2.08 M, 16 384: 2.48 -> 2.43 M; its first run at THREE QUARTERS: 2 048: 1.67 -> 1.88 M, 4 096: 2.10 -> 2.16 M, but 1 536:
  //  1.67 -> 1.60 M, 3 072: 2.01 -> 1.86 M.  Any fixed threshold moves the sawtooth, it does not remove it: what a short loop
  //  pays beyond its recording is the device work that starts when the loop ends.  The host records an encryption in 0.36 us
  //  and a run costs it ~36 us, so 2.78 M/s is the ceiling of ANY policy at 2 048 iterations: profiles/r05_short_loops.txt.)

  // NFL_HIP_EARLY_RUN=1: from 1 024 records on, every 512 records the queue asks whether the stream is idle
  // (nflhip_stream_idle: one hipStreamQuery) and runs at once if it is.  Off by default: measured on the LWE demo's
  // 2 048-iteration loop it changes nothing (profiles/r02_late_early_run.txt); results are identical either way.
  static bool early_run() {
    static const bool v = getenv("NFL_HIP_EARLY_RUN") && atoi(getenv("NFL_HIP_EARLY_RUN")) != 0;
    return v;
  }
  lazy() : rec_run_(1), done_run_(0), next_run_(std::min(min_run(), max_queue())), th(nullptr), th_pid(0), th_failed(false), st(0), w_sleeping(false), u_sleeping(false), quit(false),
           launches(0), coalesced(0), fused_fwd(0), fused_inv(0), small_(nullptr), small_cap_(0) {
    ctx_t::inst();  // (the context is constructed first, so it is destroyed last)
    alive() = true;
    queue_registry::get().add(&lazy::run_if_alive);
  }
  ~lazy() {
    alive() = false;
    if (th && th_pid == long(getpid())) {
      try {   // the run in flight holds raw pointers into this object: see it out (what it reports has nobody to go to)
        std::lock_guard<light_lock> lk(mu);
        collect();
      } catch (...) {
      }
      {
        std::lock_guard<std::mutex> lk(wm);
        quit.store(true);
        wcv.notify_all();
      }
      th->join();
      delete th;
    }
    if (small_ && ctx_t::alive()) nflhip_free(ctx_t::get(), small_);
  }
  static bool &alive() {
    static bool a = false;
    return a;
  }
  static void run_if_alive() {  // what queue_registry calls (possibly while the program's statics are being destroyed)
    if (alive() && ctx_t::alive()) inst().flush();
  }
  static lazy &inst() {
    static lazy l;
    return l;
  }
  // whether two recorded operations may share a launch: everything a launch takes from its first member
  static bool same_signature(const op &a, const op &b) {
    if (a.kind != b.kind || a.post != b.post) return false;
    if (a.kind == K_EVAL) return a.len == b.len && a.nin == b.nin && std::memcmp(a.e.code, b.e.code, a.len) == 0;
    if (a.kind == K_SAMPLE || a.kind == K_GAUSS) return a.s.dist == b.s.dist && a.s.p0 == b.s.p0 && a.s.p1 == b.s.p1 && a.s.tab == b.s.tab;
    if (a.kind == K_FILL) return a.s.sid == b.s.sid;
    if (a.kind == K_FWD_FMA)
      return a.nin == b.nin && a.f.tab == b.f.tab && a.f.amp[0] == b.f.amp[0] && a.f.amp[1] == b.f.amp[1] && a.f.amp[2] == b.f.amp[2];
    if (a.kind == K_FMA_INV) return a.e.code[0] == b.e.code[0];
    return true;
  }
  // ---- transform fusion.  Code written against the reference transforms, combines, transforms back:
  //        u.ntt_pow_phi(); e.ntt_pow_phi(); r = u * key + e;          out = rb - ra * s; out.invntt_pow_invphi();
  // (tests/nfllib_demo_main_op.cpp:26-58).  When this context runs such a sequence as ONE kernel (nflhip_has_fused_kernels),
  // a queue run rewrites what it recorded before it levels it:
  //   * K_GAUSS x, K_NTT_FWD x, K_GAUSS e, K_NTT_FWD e, K_EVAL r = x * k + e (either operand order; a second K_EVAL on the
  //     same x with its own k, e joins) -> K_FWD_FMA, provided nothing else reads the sampled or transformed x / e and
  //     their handles are gone (the queue holds the last reference): those polynomials then never exist in HBM -- the
  //     samplers write one byte per coefficient (nflhip_sample_gauss_small_seq_dev) and the kernel transforms in registers;
  //   * K_EVAL t = c +- a * b, K_NTT_INV t with nothing reading t in between -> K_FMA_INV.
  // Results are bit-identical to the operator-by-operator run.  NFL_HIP_NO_FUSION=1 switches the rewriting off.
  static bool fusion_on() {
    static const bool v = !getenv("NFL_HIP_NO_FUSION") && nflhip_has_fused_kernels(ctx_t::get()) != 0;
    return v;
  }
  // c +- a * b as a 5-byte postfix program over three distinct operands: {a, b, c, subtract}, or false.  A record that
  // carries a joined transform (op::post, join_transform below) is NOT that expression: its result is the transformed
  // value, and a rewrite that took only the expression would drop the transform.  The one place that models a joined
  // transform -- the expression followed by its own inverse transform -- asks for it by name (`joined`).
  static bool parse_fma(const op &o, int &a, int &b, int &c, bool &sub, unsigned char joined = 0) {
    if (o.kind != K_EVAL || o.len != 5 || o.nin != 3 || o.post != joined) return false;
    const unsigned char *q = o.e.code;
    if (q[0] < 3 && q[1] < 3 && q[2] == NFLHIP_EXPR_MUL && q[3] < 3 && q[4] == NFLHIP_EXPR_ADD) {            // a b * c +
      a = q[0]; b = q[1]; c = q[3]; sub = false;
    } else if (q[0] < 3 && q[1] < 3 && q[2] < 3 && q[3] == NFLHIP_EXPR_MUL && (q[4] == NFLHIP_EXPR_ADD || q[4] == NFLHIP_EXPR_SUB)) {  // c a b * +-
      c = q[0]; a = q[1]; b = q[2]; sub = q[4] == NFLHIP_EXPR_SUB;
    } else {
      return false;
    }
    return a != b && a != c && b != c;
  }
  static bool mentions(const op &o, unsigned pin) {
    if (o.kind == K_NOP) return false;
    if (o.out_pin == pin || (o.kind == K_FWD_FMA && o.f.out2 && o.f.out2_pin == pin)) return true;
    if (o.kind == K_EVAL || o.kind == K_FWD_FMA || o.kind == K_FMA_INV)
      for (int j = 0; j < o.nin; ++j)
        if (o.in_pin[j] == pin) return true;
    return false;
  }
  void *small_buffer(size_t bytes) {
    if (bytes > small_cap_) {
      nflhip_ctx *c = ctx_t::get();
      if (small_) {
        check(c, nflhip_stream_sync(c, ctx_t::queue()), "compact sampler buffer");   // (its last readers)
        nflhip_free(c, small_);
        small_ = nullptr;
        small_cap_ = 0;
      }
      size_t cap = size_t(1) << 20;
      while (cap < bytes) cap *= 2;
      check(c, nflhip_malloc(c, &small_, cap), "compact sampler buffer");
      small_cap_ = cap;
    }
    return small_;
  }
  // the narrowest compact format that holds every sample of `tab` times `amp` (NFLHIP_FMT_I8 / I16 / I32; 99 = none)
  static int small_format(const nflhip_gauss *tab, uint32_t amp) {
    struct last_t { const nflhip_gauss *tab; uint64_t mag; };
    static thread_local last_t last = {nullptr, 0};
    if (last.tab != tab) {
      long long x_min = 0;
      size_t entries = 0;
      check(ctx_t::get(), nflhip_gauss_info(tab, &x_min, &entries, nullptr, nullptr, nullptr, nullptr), "gaussian table");
      const long long hi = x_min + (long long)entries - 1;
      last.tab = tab;
      last.mag = uint64_t(std::max(x_min < 0 ? -x_min : x_min, hi < 0 ? -hi : hi));
    }
    const uint64_t v = last.mag * uint64_t(amp);
    return v <= 127 ? NFLHIP_FMT_I8 : v <= 32767 ? NFLHIP_FMT_I16 : v <= 2147483647ull ? NFLHIP_FMT_I32 : 99;
  }
  void fuse(run_t &r, std::vector<op> &ops) {
    const int n = int(ops.size());
    if (n < 2 || !fusion_on()) return;
    // definitions: prev[i] = the operation that wrote ops[i].out before i (what an in-place transform reads), def[i][j] =
    // the one that wrote input j of an expression; uses[d] = reads of the value operation d wrote; -1 = from before this run
    std::vector<int> prev(size_t(n), -1), uses(size_t(n), 0), def(size_t(n) * 3, -1);
    std::vector<int> &fw = r.fw;   // per pin: the recorded operation that last wrote the value
    fw.assign(r.held.size(), -1);
    bool any_fwd = false, any_inv = false;
    for (int i = 0; i < n; ++i) {
      op &o = ops[size_t(i)];
      if (o.kind == K_EVAL)
        for (int j = 0; j < o.nin; ++j) {
          const int d = fw[o.in_pin[j]];
          if (j < 3) def[size_t(i) * 3 + size_t(j)] = d;
          if (d >= 0) ++uses[size_t(d)];
        }
      prev[size_t(i)] = fw[o.out_pin];
      if ((o.kind == K_NTT_FWD || o.kind == K_NTT_INV) && fw[o.out_pin] >= 0) ++uses[size_t(fw[o.out_pin])];
      fw[o.out_pin] = i;
      any_fwd |= o.kind == K_NTT_FWD || o.post == K_NTT_FWD;
      any_inv |= o.kind == K_NTT_INV || o.post == K_NTT_INV;
    }
    // the value an operation wrote is still its payload's at the end of the run: only fusable away when no handle is left
    auto dead_after = [&](int d) {
      const unsigned k = ops[size_t(d)].out_pin;
      return fw[k] != d || r.dead[k] != 0;   // (this run's pin was the last reference when the run was taken: no handle, no later record)
    };
    // a sampled-and-transformed polynomial nobody else sees: -> index of its K_GAUSS record, or -1
    auto gauss_chain = [&](int dn, int want_uses) {
      if (dn >= 0 && ops[size_t(dn)].kind == K_GAUSS && ops[size_t(dn)].post == K_NTT_FWD) {   // the transform joined its constructor's record
        const op &g = ops[size_t(dn)];
        if (uses[size_t(dn)] != want_uses || !dead_after(dn) || (g.s.p1 >> 32) != 0) return -1;
        return small_format(g.s.tab, uint32_t(g.s.p1)) > NFLHIP_FMT_I32 ? -1 : dn;
      }
      if (dn < 0 || ops[size_t(dn)].kind != K_NTT_FWD || uses[size_t(dn)] != want_uses || !dead_after(dn)) return -1;
      const int g = prev[size_t(dn)];
      if (g < 0 || ops[size_t(g)].kind != K_GAUSS || ops[size_t(g)].post != 0 || uses[size_t(g)] != 1 || (ops[size_t(g)].s.p1 >> 32) != 0)
        return -1;   // (post: the constructor's record already carries one transform; this would be the second)
      if (small_format(ops[size_t(g)].s.tab, uint32_t(ops[size_t(g)].s.p1)) > NFLHIP_FMT_I32) return -1;
      return g;
    };
    if (any_inv)
      for (int i = 0; i < n; ++i) {
        op &t = ops[size_t(i)];
        if (t.kind == K_EVAL && t.post == K_NTT_INV) {   // the transform joined the expression's record: rewrite in place
          int a, b, c;
          bool sub;
          if (!parse_fma(t, a, b, c, sub, K_NTT_INV)) continue;
          pay_t *pc = t.e.in[c], *pa = t.e.in[a], *pb = t.e.in[b];
          const unsigned kc = t.in_pin[c], ka = t.in_pin[a], kb = t.in_pin[b];
          t.kind = K_FMA_INV;
          t.post = 0;
          t.nin = 3;
          t.len = 1;
          t.e.in[0] = pc;
          t.e.in[1] = pa;
          t.e.in[2] = pb;
          t.in_pin[0] = kc;
          t.in_pin[1] = ka;
          t.in_pin[2] = kb;
          t.e.code[0] = sub ? 1 : 0;
          ++fused_inv;
          continue;
        }
        if (t.kind != K_NTT_INV) continue;
        const int d = prev[size_t(i)];
        int a, b, c;
        bool sub;
        if (d < 0 || i - d > 4 || uses[size_t(d)] != 1 || !parse_fma(ops[size_t(d)], a, b, c, sub)) continue;
        op &e = ops[size_t(d)];
        bool clean = true;   // nothing between the two rewrites an operand (the fused operation reads them at i, not at d)
        for (int k = d + 1; k < i && clean; ++k)
          clean = ops[size_t(k)].kind == K_NOP || (ops[size_t(k)].out_pin != e.in_pin[0] && ops[size_t(k)].out_pin != e.in_pin[1] && ops[size_t(k)].out_pin != e.in_pin[2]);
        if (!clean) continue;
        pay_t *pc = e.e.in[c], *pa = e.e.in[a], *pb = e.e.in[b];
        t.kind = K_FMA_INV;
        t.nin = 3;
        t.len = 1;
        t.e.in[0] = pc;
        t.e.in[1] = pa;
        t.e.in[2] = pb;
        t.in_pin[0] = e.in_pin[c];
        t.in_pin[1] = e.in_pin[a];
        t.in_pin[2] = e.in_pin[b];
        t.e.code[0] = sub ? 1 : 0;
        e.kind = K_NOP;
        ++fused_inv;
      }
    if (!any_fwd) return;
    // forward: candidates per transformed x (an expression names it once; a second expression on the same x joins)
    struct cand { int i, xs, ks, es, gx, ge; };
    std::vector<cand> cands;
    for (int i = 0; i < n; ++i) {
      int a, b, c;
      bool sub;
      if (!parse_fma(ops[size_t(i)], a, b, c, sub) || sub) continue;
      for (int turn = 0; turn < 2; ++turn) {
        const int xs = turn ? b : a, ks = turn ? a : b;
        const int dx = def[size_t(i) * 3 + size_t(xs)], de = def[size_t(i) * 3 + size_t(c)];
        if (dx < 0 || de < 0 || dx == de) continue;
        const int ux = uses[size_t(dx)];
        if (ux != 1 && ux != 2) continue;
        const int gx = gauss_chain(dx, ux), ge = gauss_chain(de, 1);
        if (gx < 0 || ge < 0 || ops[size_t(gx)].s.tab != ops[size_t(ge)].s.tab) continue;
        cands.push_back(cand{i, xs, ks, c, gx, ge});
        break;
      }
    }
    for (size_t q = 0; q < cands.size(); ++q) {
      const cand &c0 = cands[q];
      if (c0.i < 0) continue;
      const int dx = def[size_t(c0.i) * 3 + size_t(c0.xs)];
      const cand *c1 = nullptr;
      if (uses[size_t(dx)] == 2) {   // the other reader of NTT(x) must be a candidate too, close by, and independent of this one
        for (size_t r = q + 1; r < cands.size() && !c1; ++r)
          if (cands[r].i >= 0 && def[size_t(cands[r].i) * 3 + size_t(cands[r].xs)] == dx) c1 = &cands[r];
        if (!c1 || c1->i - c0.i > 4) continue;
        const op &e0 = ops[size_t(c0.i)], &e1 = ops[size_t(c1->i)];
        bool clean = e1.in_pin[c1->ks] != e0.out_pin && e1.out_pin != e0.out_pin;   // (the fused operation writes both results at e1's place)
        for (int k = c0.i + 1; k < c1->i && clean; ++k)
          clean = !mentions(ops[size_t(k)], e0.out_pin) && (ops[size_t(k)].kind == K_NOP || ops[size_t(k)].out_pin != e0.in_pin[c0.ks]);
        if (!clean) continue;
      }
      const op e0 = ops[size_t(c0.i)];
      op &t = ops[size_t(c1 ? c1->i : c0.i)];
      const op e1 = t;
      const op &gx = ops[size_t(c0.gx)], &g0 = ops[size_t(c0.ge)];
      t.kind = K_FWD_FMA;
      t.out = e0.out;
      t.out_pin = e0.out_pin;
      t.nin = c1 ? 2 : 1;
      t.len = 0;
      t.f.in[0] = e0.e.in[c0.ks];
      t.f.in[1] = c1 ? e1.e.in[c1->ks] : nullptr;
      t.f.in[2] = t.f.in[3] = nullptr;
      t.in_pin[0] = e0.in_pin[c0.ks];
      t.in_pin[1] = c1 ? e1.in_pin[c1->ks] : 0;
      t.f.out2 = c1 ? e1.out : nullptr;
      t.f.out2_pin = c1 ? e1.out_pin : 0;
      t.f.tab = gx.s.tab;
      t.f.sid[0] = gx.s.sid;
      t.f.amp[0] = uint32_t(gx.s.p1);
      t.f.sid[1] = g0.s.sid;
      t.f.amp[1] = uint32_t(g0.s.p1);
      t.f.sid[2] = c1 ? ops[size_t(c1->ge)].s.sid : 0;
      t.f.amp[2] = c1 ? uint32_t(ops[size_t(c1->ge)].s.p1) : 0;
      // the records the fused operation stands for
      const int gone[] = {c0.gx, dx, c0.ge, def[size_t(c0.i) * 3 + size_t(c0.es)], c1 ? c0.i : -1, c1 ? c1->ge : -1,
                          c1 ? def[size_t(c1->i) * 3 + size_t(c1->es)] : -1};
      for (int g : gone)
        if (g >= 0) ops[size_t(g)].kind = K_NOP;
      if (c1) const_cast<cand *>(c1)->i = -1;
      ++fused_fwd;
    }
  }
  // whether this ring's operations can be deferred at all: dense chunks, vectors of 16 bytes, sequence samplers
  static bool usable() {
    return deferred_flag().load(std::memory_order_relaxed) && ctx_t::chunk_bytes == ctx_t::poly_bytes && P::degree >= 8 &&
           P::degree * sizeof(T) >= 16;
  }
  // ascending order for addresses that usually are `period` interleaved ascending sequences already (a loop body that
  // transforms u, e1, e2 -- each kind a dense array of its own -- yields u0 e1_0 e2_0 u1 e1_1 e2_1 ...): merged in O(n)
  static void sort_interleaved(std::vector<char *> &v) {
    if (std::is_sorted(v.begin(), v.end())) return;
    for (size_t period = 2; period <= 8 && period * 2 <= v.size(); ++period) {
      bool ok = true;
      for (size_t i = period; i < v.size() && ok; ++i) ok = !(v[i] < v[i - period]);
      if (!ok) continue;
      std::vector<char *> out;
      out.reserve(v.size());
      for (size_t r = 0; r < period; ++r) {
        const size_t mid = out.size();
        for (size_t i = r; i < v.size(); i += period) out.push_back(v[i]);
        std::inplace_merge(out.begin(), out.begin() + ptrdiff_t(mid), out.end());
      }
      v.swap(out);
      return;
    }
    std::sort(v.begin(), v.end());
  }
  // the recording run's reference to a payload, taken the first time one of its operations mentions it (the recording scratch
  // is tagged with the run at the same moment: "mentioned in this run" and "pinned by this run" are one fact)
  pay_t *pin(pay_t *p) {
    if (p->rec_run != rec_run_) {
      p->rec_run = rec_run_;
      p->rec_w = p->rec_r = -1;
      p->pin_at = unsigned(pins.size());
      pins.push_back(p->shared_from_this());
      ++p->qrefs;
    }
    return p;
  }
  // A transform recorded on a value that a Gaussian constructor or an expression of THIS recording run produced and that
  // nothing has read since does not become a record of its own: the producing operation notes "then transform in place"
  // (op::post).  The reference's loops are written that way -- poly_p u{gaussian}; u.ntt_pow_phi();  out = rb - ra * s;
  // out.invntt_pow_invphi(); -- and every record costs the host the same whatever it stands for.  Results are those of the
  // separate records; NFL_HIP_NO_FUSION=1 switches this off together with the transform fusion.
  static bool joining_on() {
    static const bool v = !getenv("NFL_HIP_NO_FUSION");
    return v;
  }
  bool join_transform(pay_t *p, int kind) {
    if (!joining_on()) return false;
    std::lock_guard<light_lock> lk(mu);
    if (p->rec_run != rec_run_ || p->rec_w < 0 || p->rec_r > p->rec_w || p->poisoned) return false;
    op &t = q[size_t(p->rec_w)];
    if (t.post || t.out != p || !((t.kind == K_GAUSS && kind == K_NTT_FWD) || t.kind == K_EVAL)) return false;
    t.post = static_cast<unsigned char>(kind);
    return true;
  }
  // `fill(op &)` writes the record in place, in the queue; the payloads it names are pinned here
  template <class F> void record(F fill) {
    std::lock_guard<light_lock> lk(mu);
    q.emplace_back();
    op &o = q.back();
#if defined(__x86_64__)
    // the vector's storage was last read by the queue's thread (another core, often another L3): ask for the lines this loop will
    // write a dozen records from now in exclusive state NOW, so that the stores do not each wait for the invalidation
    if (q.size() + 16 <= q.capacity()) {
      const char *ahead = reinterpret_cast<const char *>(&o + 16);
      __asm__ volatile("prefetchw %0\n\tprefetchw %1" : : "m"(*ahead), "m"(*(ahead + 64)));
    }
#endif
    o.nin = 0;
    o.len = 0;
    o.post = 0;
    try {  // inputs must hold a device value (or be produced by the queue) before the operation counts as recorded
      fill(o);
      for (int j = 0; j < o.nin; ++j) o.e.in[j]->dev_ro_nf();
      if (o.kind == K_NTT_FWD || o.kind == K_NTT_INV) o.out->dev_ro_nf();
    } catch (...) {
      q.pop_back();
      throw;
    }
    const int at = int(q.size()) - 1;
    for (int j = 0; j < o.nin; ++j) {
      pin(o.e.in[j])->rec_r = at;
      o.in_pin[j] = o.e.in[j]->pin_at;
    }
    pin(o.out)->rec_w = at;
    o.out_pin = o.out->pin_at;
    o.out->qrun = rec_run_;
    o.out->dev_valid = true;
    o.out->host_valid = false;
    if (o.kind != K_NTT_FWD && o.kind != K_NTT_INV) o.out->poisoned = false;  // overwritten entirely
    const size_t n = q.size();
    if (!threaded()) {
      if (n >= max_queue()) hand_over();
    } else if (n >= next_run_ && (n >= 2 * max_queue() || (n % 64 == 0 && st.load(std::memory_order_relaxed) != 1)) && have_thread()) {
      // Run lengths of a loop grow geometrically (min_run(), twice that, ... up to max_queue()): the device starts on the loop's
      // first few hundred iterations while the host is still recording, and the later runs are long enough for the device's
      // best rate (per run it pays three sampler launches and the drain of the fused kernel, whatever the length).  A run waits
      // for the queue's thread only at twice the full length; flush() -- somebody waited: the loop is over -- starts anew.
      hand_over();
      next_run_ = std::min(next_run_ * 2, max_queue());
    } else if (n >= 2 * max_queue()) {
      hand_over();   // (no thread after all)
    } else if (early_run() && n >= 1024 && n % 512 == 0) {
      // a loop shorter than the queue: do not let the device sit idle until the loop's end
      int idle = 0;
      if (nflhip_stream_idle(ctx_t::get(), ctx_t::queue(), &idle) == NFLHIP_OK && idle) hand_over();
    }
  }
  // everything recorded so far has been launched (or has failed: rethrown here) when this returns
  void flush() {
    std::lock_guard<light_lock> lk(mu);
    next_run_ = std::min(min_run(), max_queue());
    collect();
    if (q.empty()) return;
    take();
    execute(fly);   // somebody waits for it: no hand-over
    retire();
  }
  // the records so far become the run in flight (the previous one is retired first): on the queue's thread if there is one
  void hand_over() {
    collect();
    if (q.empty()) return;
    take(clean_cut());
    if (have_thread()) {
      fly.remote = true;
      post();
    } else {
      execute(fly);
      retire();
    }
  }
  bool have_thread() {   // (under mu)
    if (!threaded() || th_failed) return false;
    const long pid = long(getpid());
    if (th && th_pid == pid) return true;
    if (th) return false;   // a forked child: the thread stayed with the parent; runs are executed by their callers here
    try {
      th_pid = pid;
      th = new std::thread([this] { worker(); });
    } catch (...) {
      th = nullptr;
      th_failed = true;
      return false;
    }
    return true;
  }
  // Where to cut the queue when a run starts by itself in the middle of a loop.  The iteration that is being recorded right now has
  // sampled its temporaries but not consumed them yet (poly_p u{gaussian}; u.ntt_pow_phi(); ra = u * pka + e1; | rb = u * pkb + e2;):
  // a run that takes its first half cannot fuse it -- u has a handle and another reader to come -- and launches nine small
  // kernels for that one iteration instead (a fifth of the device's time per run, rocprofv3 timeline of the LWE loop).  So the
  // records from the first one that mentions a Gaussian temporary of the last few records whose HANDLE IS STILL ALIVE stay in
  // the queue for the next run.  -> number of records the run takes (all of them when there is no such temporary).
  size_t clean_cut() const {
    const size_t n = q.size(), window = 32;
    const size_t from = n > window ? n - window : 0;
    unsigned open_pin[8];
    int nopen = 0;
    for (size_t i = from; i < n && nopen < 8; ++i) {
      const op &o = q[i];
      if (o.kind == K_GAUSS && pins[o.out_pin].use_count() - o.out->qrefs >= 1) open_pin[nopen++] = o.out_pin;
    }
    if (!nopen) return n;
    for (size_t i = from; i < n; ++i)
      for (int k = 0; k < nopen; ++k)
        if (mentions(q[i], open_pin[k])) return i ? i : n;   // (nothing in front of it: take everything)
    return n;
  }
  void take(size_t cut = size_t(-1)) {   // (under mu, nothing in flight); records from `cut` on stay in the queue
    op rest[32];
    size_t nrest = 0;
    if (cut < q.size()) {
      nrest = q.size() - cut;
      std::copy(q.begin() + ptrdiff_t(cut), q.end(), rest);
      q.resize(cut);
    }
    fly.ops.swap(q);
    fly.held.swap(pins);
    if (q.capacity() < fly.ops.capacity()) q.reserve(fly.ops.capacity());
    fly.id = rec_run_++;   // (what is recorded from now on cannot join operations of this run, and pins again)
    for (size_t i = 0; i < nrest; ++i) {   // the records that stay: recorded again, in the new recording run
      q.push_back(rest[i]);
      op &o = q.back();
      const int at = int(q.size()) - 1;
      const bool two = o.kind == K_FWD_FMA && o.f.out2;   // (never recorded; kept for completeness)
      for (int j = 0; j < o.nin; ++j) {
        pin(o.e.in[j])->rec_r = at;
        o.in_pin[j] = o.e.in[j]->pin_at;
      }
      pin(o.out)->rec_w = at;
      o.out_pin = o.out->pin_at;
      o.out->qrun = rec_run_;
      if (two) {
        pin(o.f.out2)->rec_w = at;
        o.f.out2_pin = o.f.out2->pin_at;
        o.f.out2->qrun = rec_run_;
      }
    }
    fly.launched.assign(fly.ops.size(), 0);
    const size_t np = fly.held.size();
    fly.dev.resize(np);
    fly.dead.resize(np);
    for (size_t k = 0; k < np; ++k) {
      fly.dev[k] = fly.held[k]->dev;
      fly.dead[k] = fly.held[k].use_count() == 1;
    }
    fly.complete = false;
    fly.error = nullptr;
    fly.remote = false;
    detail::sampler::get().copy_key(fly.key);
  }
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void post() {
    st.store(1, std::memory_order_seq_cst);
    if (w_sleeping.load(std::memory_order_seq_cst)) {
      std::lock_guard<std::mutex> lk(wm);
      wcv.notify_one();
    }
  }
  void worker() {
#if defined(__linux__) && defined(__GLIBC__)
    pthread_setname_np(pthread_self(), "nflhip-queue");
#endif
    for (;;) {
      unsigned spins = 0;
      while (st.load(std::memory_order_acquire) != 1 && !quit.load(std::memory_order_relaxed)) {
        if (++spins < 20000) {   // (a loop hands over the next run within a fraction of a millisecond: worth a short spin)
          cpu_relax();
          continue;
        }
        std::unique_lock<std::mutex> lk(wm);
        w_sleeping.store(true, std::memory_order_seq_cst);
        wcv.wait(lk, [this] { return st.load(std::memory_order_seq_cst) == 1 || quit.load(); });
        w_sleeping.store(false, std::memory_order_seq_cst);
      }
      if (st.load(std::memory_order_acquire) != 1) return;   // quit
      execute(fly);
      st.store(2, std::memory_order_seq_cst);
      if (u_sleeping.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lk(wm);
        ucv.notify_all();
      }
    }
  }
  // wait for the run in flight, if any, and retire it (under mu)
  void collect() {
    int s = st.load(std::memory_order_acquire);
    if (s == 0) return;
    if (s == 1 && th_pid != long(getpid())) {   // a forked child: the run in flight stayed with the parent's thread
      fly.complete = false;
      fly.error = std::make_exception_ptr(std::runtime_error("nfl(hip): the process forked while a queue run was in flight"));
      retire();
      return;
    }
    for (unsigned spins = 0; s != 2; s = st.load(std::memory_order_acquire)) {
      if (++spins < 20000) {
        cpu_relax();
        continue;
      }
      std::unique_lock<std::mutex> lk(wm);
      u_sleeping.store(true, std::memory_order_seq_cst);
      ucv.wait(lk, [this] { return st.load(std::memory_order_seq_cst) == 2; });
      u_sleeping.store(false, std::memory_order_seq_cst);
    }
    retire();
  }
  // the run in flight is over (under mu; executed here, or by the queue's thread which has set st = 2): its values stop being
  // queued, its references go; if a launch failed, what was never launched holds no value -- later accesses throw
  // (payload::usable) -- and so does everything recorded since (it was recorded on top of values that do not exist): rethrown
  void retire() {
    st.store(0, std::memory_order_relaxed);
    done_run_.store(fly.id, std::memory_order_relaxed);
    const bool failed = !fly.complete;
    if (failed)
      for (size_t i = 0; i < fly.ops.size(); ++i)
        if (!fly.launched[i]) {
          fly.ops[i].out->dev_valid = false;
          fly.ops[i].out->poisoned = true;
        }
    fly.ops.clear();
    for (size_t k = 0; k < fly.held.size(); ++k) {
      pay_t *p = fly.held[k].get();
      p->dev = fly.dev[k];   // (results that had no buffer got theirs from the run)
      --p->qrefs;
    }
    if (failed && !q.empty()) {
      for (auto &o : q) {
        o.out->dev_valid = false;
        o.out->poisoned = true;
      }
      q.clear();
      for (auto &p : pins) --p->qrefs;
      fly.held.insert(fly.held.end(), pins.begin(), pins.end());
      pins.clear();
      done_run_.store(rec_run_++, std::memory_order_relaxed);
    }
    if (ctx_t::alive()) {   // the temporaries' buffers go back to the pool one by one: its lock is taken once for all of them
      std::lock_guard<light_lock> pool(ctx_t::inst().mu);
      fly.held.clear();
    } else {
      fly.held.clear();
    }
    if (failed) {
      std::exception_ptr e = fly.error;
      fly.error = nullptr;
      if (e) std::rethrow_exception(e);
      throw std::runtime_error("nfl(hip): a deferred operation failed");
    }
  }
  // One queue run: fuse, level, group, launch.  Runs on the queue's thread or on the thread that waits for it; touches nothing
  // of the recording state (see payload: `dev` of results without a buffer, and the levelling scratch).  Never throws: what
  // stops it is kept in r.error, r.launched says which operations were issued.
  void execute(run_t &r) {
    static const bool stats = getenv("NFL_HIP_TRACE_DEFERRED") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t nops = r.ops.size();
    try {
      execute_body(r);
      r.complete = true;
    } catch (...) {
      r.error = std::current_exception();
    }
    if (stats) {
      const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
      std::fprintf(stderr, "nfl(hip) queue run: %zu records in %.1f us (%s)\n", nops, us, r.remote ? "the queue's thread" : "the calling thread");
    }
  }
  void execute_body(run_t &r) {
    // The queue's thread works on a COPY of the records: it rewrites them (fusion), and lines it has written would have to come
    // back from its cache, one by one, when the recording thread fills the same vector again two runs later.
    if (r.remote) r.work.assign(r.ops.begin(), r.ops.end());
    std::vector<op> &ops = r.remote ? r.work : r.ops;
    std::vector<unsigned char> &launched = r.launched;
    std::vector<void *> &D = r.dev;   // per pin: the value's buffer
    fuse(r, ops);
    // ---- 1. levels (per pin: the last level that writes / reads the value)
    std::vector<int> &wlev = r.wlev, &rlev = r.rlev;
    wlev.assign(r.held.size(), -1);
    rlev.assign(r.held.size(), -1);
    std::vector<int> lvl(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); ++i) {
      op &o = ops[i];
      if (o.kind == K_NOP) {   // its work moved into a fused operation
        lvl[i] = -1;
        launched[i] = 1;
        continue;
      }
      int L = 0;
      for (int j = 0; j < o.nin; ++j) L = std::max(L, wlev[o.in_pin[j]] + 1);
      L = std::max(L, std::max(wlev[o.out_pin], rlev[o.out_pin]) + 1);
      const bool two_results = o.kind == K_FWD_FMA && o.f.out2;
      if (two_results) L = std::max(L, std::max(wlev[o.f.out2_pin], rlev[o.f.out2_pin]) + 1);
      lvl[i] = L;
      wlev[o.out_pin] = L;
      if (two_results) wlev[o.f.out2_pin] = L;
      for (int j = 0; j < o.nin; ++j) rlev[o.in_pin[j]] = std::max(rlev[o.in_pin[j]], L);
      if (o.kind == K_NTT_FWD || o.kind == K_NTT_INV) rlev[o.out_pin] = std::max(rlev[o.out_pin], L);
    }
    // ---- 2. groups: (level, signature) -> operations in program order.  A loop produces a handful of distinct
    // signatures, so a linear table of the ones seen (64-bit FNV-1a of the fields, plus the level) beats a map.
    struct gkey { int level; uint64_t hash; };
    std::vector<gkey> keys;
    std::vector<std::vector<size_t>> members;
    auto mix = [](uint64_t h, uint64_t v) {  // (one multiply-xorshift round per 64-bit field: the signatures are a few words)
      h = (h ^ v) * 0x9E3779B97F4A7C15ull;
      return h ^ (h >> 29);
    };
    static_assert(NFLHIP_EXPR_MAX_LEN <= 24, "the program is hashed as three words");
    for (size_t i = 0; i < ops.size(); ++i) {
      const op &o = ops[i];
      if (o.kind == K_NOP) continue;
      uint64_t h = mix(0xcbf29ce484222325ull, uint64_t(o.kind) | (uint64_t(o.post) << 8));
      if (o.kind == K_FWD_FMA) {
        h = mix(mix(mix(h, uint64_t(reinterpret_cast<uintptr_t>(o.f.tab))), (uint64_t(o.f.amp[0]) << 32) | o.f.amp[1]), (uint64_t(o.f.amp[2]) << 8) | o.nin);
      } else if (o.kind == K_FMA_INV) {
        h = mix(h, o.e.code[0]);
      } else if (o.kind == K_EVAL) {
        uint64_t w[3] = {0, 0, 0};
        std::memcpy(w, o.e.code, size_t(o.len));
        h = mix(mix(mix(mix(h, w[0]), w[1]), w[2]), (uint64_t(o.len) << 8) | uint64_t(o.nin));
      } else if (o.kind == K_SAMPLE || o.kind == K_GAUSS || o.kind == K_FILL) {
        h = mix(mix(mix(mix(h, uint64_t(o.s.dist)), o.s.p0), o.s.p1), uint64_t(reinterpret_cast<uintptr_t>(o.s.tab)));
        if (o.kind == K_FILL) h = mix(h, o.s.sid);
      }
      size_t g = keys.size();
      for (size_t k = keys.size(); k-- > 0;)   // (recent groups first: neighbouring operations repeat)
        if (keys[k].level == lvl[i] && keys[k].hash == h && same_signature(ops[members[k][0]], o)) { g = k; break; }
      if (g == keys.size()) {
        keys.push_back(gkey{lvl[i], h});
        members.emplace_back();
        members.back().reserve(ops.size() / 4 + 1);
      }
      members[g].push_back(i);
    }
    // groups run level by level (inside a level the order is irrelevant: they are independent)
    std::vector<size_t> order(keys.size());
    for (size_t g = 0; g < order.size(); ++g) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return keys[x].level < keys[y].level; });
    nflhip_ctx *ctx = ctx_t::get();
    void *st = ctx_t::queue();
    struct { const unsigned char *key; } smp = {r.key};  // the key as it was when the run was taken
    static const bool trace = getenv("NFL_HIP_TRACE_DEFERRED") != nullptr;
    // in-place transforms of the results of `idx` (mutually independent): by address, so that neighbours become one dense batch
    auto launch_transforms = [&](const std::vector<size_t> &idx, int tkind) {
      std::vector<char *> ptr;
      ptr.reserve(idx.size());
      for (size_t i : idx) ptr.push_back(static_cast<char *>(D[ops[i].out_pin]));
      sort_interleaved(ptr);
      for (size_t a = 0; a < ptr.size();) {
        size_t b = a + 1;
        while (b < ptr.size() && ptr[b] == ptr[b - 1] + ctx_t::chunk_bytes) ++b;
        check(ctx, tkind == K_NTT_FWD ? nflhip_ntt_fwd_dev(ctx, ptr[a], b - a, st) : nflhip_ntt_inv_dev(ctx, ptr[a], b - a, st),
              "deferred transform");
        ++launches;
        coalesced += b - a;
        a = b;
      }
    };
    // (operations that carry a joined transform count as launched only once it has been issued)
    auto finish_post = [&](const std::vector<size_t> &idx) {
      const int post = ops[idx[0]].post;
      if (!post) return;
      for (size_t i : idx) launched[i] = 0;
      launch_transforms(idx, post);
      for (size_t i : idx) launched[i] = 1;
    };
    for (size_t gi : order) {
      std::vector<size_t> &idx = members[gi];
      const int kind = ops[idx[0]].kind;
      const size_t launches_before = launches;
      struct tracer {
        bool on; int level, kind; size_t n; const std::atomic<size_t> &now; size_t before;
        ~tracer() { if (on) std::fprintf(stderr, "nfl(hip) deferred: level %d kind %d: %zu operations -> %zu launches\n", level, kind, n, now.load() - before); }
      } tr{trace, keys[gi].level, kind, idx.size(), launches, launches_before};
      if ((kind == K_SAMPLE || kind == K_GAUSS) && idx.size() >= 4) {
        // A loop body that draws several polynomials of one distribution (e1, e2 of an encryption) interleaves their
        // stream ids: k+1, k+2, k+4, k+5, ...  Find the period of the id differences and regroup the operations into
        // that many arithmetic progressions, each of which then gets its own dense array of buffers and is one launch.
        for (size_t period = 2; period <= 8 && period * 2 <= idx.size(); ++period) {
          bool periodic = true, constant = true;
          for (size_t i = 0; i + 1 < idx.size() && periodic; ++i) {
            const uint64_t d = ops[idx[i + 1]].s.sid - ops[idx[i]].s.sid;
            if (i + 1 + period < idx.size()) periodic = d == ops[idx[i + 1 + period]].s.sid - ops[idx[i + period]].s.sid;
            constant &= d == ops[idx[1]].s.sid - ops[idx[0]].s.sid;
          }
          if (constant) break;
          if (periodic) {
            std::vector<size_t> re;
            for (size_t r = 0; r < period; ++r)
              for (size_t i = r; i < idx.size(); i += period) re.push_back(idx[i]);
            idx.swap(re);
            break;
          }
        }
      }
      // ---- 3. buffers for results that have none yet: consecutive, in program order
      std::vector<size_t> need;
      for (size_t i : idx)
        if (!D[ops[i].out_pin]) need.push_back(i);
      if (!need.empty()) {
        std::vector<void *> bufs(need.size());
        ctx_t::acquire_many(need.size(), bufs.data());
        for (size_t k = 0; k < need.size(); ++k) D[ops[need[k]].out_pin] = bufs[k];
      }
      if (kind == K_FWD_FMA) {   // the second results: a dense array of their own
        need.clear();
        for (size_t i : idx)
          if (ops[i].f.out2 && !D[ops[i].f.out2_pin]) need.push_back(i);
        if (!need.empty()) {
          std::vector<void *> bufs(need.size());
          ctx_t::acquire_many(need.size(), bufs.data());
          for (size_t k = 0; k < need.size(); ++k) D[ops[need[k]].f.out2_pin] = bufs[k];
        }
      }
      if (kind == K_NTT_FWD || kind == K_NTT_INV) {
        launch_transforms(idx, kind);
        for (size_t i : idx) launched[i] = 1;
        continue;
      }
      if (kind == K_FILL) {
        for (size_t i : idx) {
          check(ctx, nflhip_fill_uniform_dev(ctx, D[ops[i].out_pin], 0, 1, ops[i].s.sid, 0, st), "deferred set(uniform)");
          launched[i] = 1;
          ++launches;
          ++coalesced;
        }
        continue;
      }
      if (kind == K_SAMPLE || kind == K_GAUSS) {
        // program order; a run = consecutive buffers + stream ids in arithmetic progression
        for (size_t a = 0; a < idx.size();) {
          const op &o0 = ops[idx[a]];
          size_t b = a + 1;
          uint64_t stride = 0;
          while (b < idx.size()) {
            const op &prev = ops[idx[b - 1]], &cur = ops[idx[b]];
            if (static_cast<char *>(D[cur.out_pin]) != static_cast<char *>(D[prev.out_pin]) + ctx_t::chunk_bytes) break;
            const uint64_t d = cur.s.sid - prev.s.sid;
            if (b == a + 1) stride = d;
            else if (d != stride) break;
            ++b;
          }
          const size_t cnt = b - a;
          if (kind == K_SAMPLE)
            check(ctx, cnt == 1 ? nflhip_sample_dev(ctx, D[o0.out_pin], 0, 1, o0.s.dist, o0.s.p0, o0.s.p1, smp.key, o0.s.sid, st)
                                : nflhip_sample_seq_dev(ctx, D[o0.out_pin], cnt, o0.s.dist, o0.s.p0, o0.s.p1, smp.key, o0.s.sid, stride, st),
                  "deferred random constructor");
          else
            check(ctx, cnt == 1 ? nflhip_sample_gauss_dev(ctx, D[o0.out_pin], 0, 1, o0.s.tab, o0.s.p1, smp.key, o0.s.sid, st)
                                : nflhip_sample_gauss_seq_dev(ctx, D[o0.out_pin], cnt, o0.s.tab, o0.s.p1, smp.key, o0.s.sid, stride, st),
                  "deferred set(gaussian)");
          for (size_t k = a; k < b; ++k) launched[idx[k]] = 1;
          ++launches;
          coalesced += cnt;
          a = b;
        }
        finish_post(idx);
        continue;
      }
      // ---- K_EVAL (and the fused kinds, whose operands sit in the same slots): operands that are one polynomial for
      // (almost) the whole group split it; then stride runs
      const int nin = ops[idx[0]].nin;
      // a "key" slot holds one of a few polynomials throughout the group (at most 8, and at most every eighth operation a
      // new one); each combination of keys becomes its own sub-group, whose other operands then advance by strides
      struct subgroup { unsigned key[NFLHIP_EXPR_MAX_OPERANDS]; std::vector<size_t> idx; };   // (keys by pin; ~0u: not a key slot)
      std::vector<subgroup> sub;
      {
        bool keyslot[NFLHIP_EXPR_MAX_OPERANDS];
        const size_t cap = std::min<size_t>(idx.size() / 8 + 1, 8);
        for (int j = 0; j < nin; ++j) {
          unsigned seen[8];
          size_t ns = 0;
          bool few = idx.size() >= 2;
          for (size_t i : idx) {
            if (!few) break;
            const unsigned p = ops[i].in_pin[j];
            size_t k = ns;
            while (k-- > 0 && seen[k] != p) {}
            if (k == size_t(-1)) {
              if (ns == cap) few = false;
              else seen[ns++] = p;
            }
          }
          keyslot[j] = few && ns < idx.size();
        }
        for (size_t i : idx) {
          unsigned key[NFLHIP_EXPR_MAX_OPERANDS];
          for (int j = 0; j < nin; ++j) key[j] = keyslot[j] ? ops[i].in_pin[j] : ~0u;
          size_t g = sub.size();
          for (size_t k = sub.size(); k-- > 0;)
            if (std::equal(key, key + nin, sub[k].key)) { g = k; break; }
          if (g == sub.size()) {
            sub.emplace_back();
            std::copy(key, key + nin, sub.back().key);
          }
          sub[g].idx.push_back(i);
        }
      }
      for (auto &sv : sub) {
        std::vector<size_t> &sidx = sv.idx;
        if (kind == K_FWD_FMA) {
          // program order; a run = consecutive result buffers (both results), keys at constant strides, stream ids of
          // every Gaussian operand in arithmetic progression: the samplers' compact outputs and ONE fused launch
          const bool two = nin == 2;
          const int nx = two ? 3 : 2;
          for (size_t a = 0; a < sidx.size();) {
            const op &o0 = ops[sidx[a]];
            size_t kstride[2] = {0, 0};
            uint64_t sstride[3] = {0, 0, 0};
            size_t b = a + 1;
            while (b < sidx.size()) {
              const op &prev = ops[sidx[b - 1]], &cur = ops[sidx[b]];
              bool ok = static_cast<char *>(D[cur.out_pin]) == static_cast<char *>(D[prev.out_pin]) + ctx_t::chunk_bytes &&
                        (!two || static_cast<char *>(D[cur.f.out2_pin]) == static_cast<char *>(D[prev.f.out2_pin]) + ctx_t::chunk_bytes);
              for (int j = 0; j < nin && ok; ++j) {
                const ptrdiff_t d = static_cast<char *>(D[cur.in_pin[j]]) - static_cast<char *>(D[prev.in_pin[j]]);
                if (d < 0 || d % ptrdiff_t(ctx_t::chunk_bytes)) ok = false;
                else if (b == a + 1) kstride[j] = size_t(d) / ctx_t::chunk_bytes;
                else if (size_t(d) != kstride[j] * ctx_t::chunk_bytes) ok = false;
              }
              for (int j = 0; j < nx && ok; ++j) {
                const uint64_t d = cur.f.sid[j] - prev.f.sid[j];
                if (b == a + 1) sstride[j] = d;
                else if (d != sstride[j]) ok = false;
              }
              if (!ok) break;
              ++b;
            }
            const size_t cnt = b - a;
            int fmt = NFLHIP_FMT_I8;
            for (int j = 0; j < nx; ++j) fmt = std::max(fmt, small_format(o0.f.tab, o0.f.amp[j]));
            const size_t es = fmt == NFLHIP_FMT_I8 ? 1 : fmt == NFLHIP_FMT_I16 ? 2 : 4, each = (cnt * P::degree * es + 255) / 256 * 256;
            char *buf = static_cast<char *>(small_buffer(each * size_t(nx)));
            nflhip_operand x[3], k[2];
            void *dst[3];
            uint64_t amp[3];
            for (int j = 0; j < nx; ++j) {
              dst[j] = buf + each * size_t(j);
              amp[j] = o0.f.amp[j];
              x[j].ptr = dst[j];
              x[j].stride = 1;
              x[j].format = fmt;
            }
            // (x, e0, e1 of a fused record are draws of ONE table: one launch for the three)
            check(ctx, nflhip_sample_gauss_small_multi_dev(ctx, dst, size_t(nx), fmt, cnt, o0.f.tab, amp, smp.key, o0.f.sid, sstride, st),
                  "deferred set(gaussian), compact");
            ++launches;
            for (int j = 0; j < nin; ++j) {
              k[j].ptr = D[o0.in_pin[j]];
              k[j].stride = cnt > 1 ? kstride[j] : 0;
              k[j].format = NFLHIP_FMT_WORDS;
            }
            check(ctx, two ? nflhip_fwd_fma2_dev(ctx, D[o0.out_pin], D[o0.f.out2_pin], &x[0], &k[0], &x[1], &k[1], &x[2], cnt, st)
                           : nflhip_fwd_fma_dev(ctx, D[o0.out_pin], &x[0], &k[0], &x[1], cnt, st),
                  "deferred transform + multiply-add");
            for (size_t q = a; q < b; ++q) launched[sidx[q]] = 1;
            ++launches;
            coalesced += cnt * (two ? 8 : 5);   // (the operations the run's members were recorded as)
            a = b;
          }
          continue;
        }
        {  // by destination address (program order among equals); a loop's results already are in that order
          bool sorted = true;
          for (size_t k = 1; k < sidx.size() && sorted; ++k) sorted = !(D[ops[sidx[k]].out_pin] < D[ops[sidx[k - 1]].out_pin]);
          if (!sorted) std::stable_sort(sidx.begin(), sidx.end(), [&](size_t x, size_t y) { return D[ops[x].out_pin] < D[ops[y].out_pin]; });
        }
        for (size_t a = 0; a < sidx.size();) {
          const op &o0 = ops[sidx[a]];
          size_t stride[NFLHIP_EXPR_MAX_OPERANDS], ostride = 1;
          size_t b = a + 1;
          while (b < sidx.size()) {
            const op &prev = ops[sidx[b - 1]], &cur = ops[sidx[b]];
            bool ok = true;
            const ptrdiff_t od = static_cast<char *>(D[cur.out_pin]) - static_cast<char *>(D[prev.out_pin]);
            if (od <= 0 || od % ptrdiff_t(ctx_t::chunk_bytes)) break;
            if (kind == K_FMA_INV && size_t(od) != ctx_t::chunk_bytes) break;   // (the fused entry writes dense results)
            if (b == a + 1) ostride = size_t(od) / ctx_t::chunk_bytes;
            else if (size_t(od) != ostride * ctx_t::chunk_bytes) break;
            for (int j = 0; j < nin && ok; ++j) {
              const ptrdiff_t d = static_cast<char *>(D[cur.in_pin[j]]) - static_cast<char *>(D[prev.in_pin[j]]);
              if (d < 0 || d % ptrdiff_t(ctx_t::chunk_bytes)) ok = false;
              else if (b == a + 1) stride[j] = size_t(d) / ctx_t::chunk_bytes;
              else if (size_t(d) != stride[j] * ctx_t::chunk_bytes) ok = false;
            }
            if (!ok) break;
            ++b;
          }
          const size_t cnt = b - a;
          const void *d[NFLHIP_EXPR_MAX_OPERANDS];
          for (int j = 0; j < nin; ++j) d[j] = D[o0.in_pin[j]];
          if (kind == K_FMA_INV) {   // in[0] +- in[1] * in[2], then the inverse transform: one launch
            nflhip_operand w[3];
            for (int j = 0; j < 3; ++j) {
              w[j].ptr = d[j];
              w[j].stride = cnt > 1 ? stride[j] : 0;
              w[j].format = NFLHIP_FMT_WORDS;
            }
            check(ctx, nflhip_fma_inv_dev(ctx, D[o0.out_pin], &w[1], &w[2], &w[0], o0.e.code[0], cnt, st), "deferred multiply-add + inverse transform");
            coalesced += cnt;   // (two recorded operations per member)
          } else if (cnt == 1) {
            check(ctx, nflhip_eval_dev(ctx, D[o0.out_pin], d, size_t(nin), o0.e.code, size_t(o0.len), 1, st), "deferred operator=(expr)");
          } else {
            check(ctx, nflhip_eval_strided_dev(ctx, D[o0.out_pin], ostride, d, stride, size_t(nin), o0.e.code, size_t(o0.len), cnt, st),
                  "deferred operator=(expr)");
          }
          for (size_t k = a; k < b; ++k) launched[sidx[k]] = 1;
          ++launches;
          coalesced += cnt;
          a = b;
        }
      }
      finish_post(idx);
    }
  }
};
}  // namespace detail
}  // namespace nfl
#endif  // NFL_HIP_QUEUE_HPP
