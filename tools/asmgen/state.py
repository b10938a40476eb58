"""The register map and the generator's configuration: every module-level value of the generator lives HERE, and the other
modules read it as `cfg.NAME` at the moment they emit -- configure() (and main(), between kernel families) rebinds these names, so a
`from .state import NAME` would freeze the value of import time.  Nothing in here emits an instruction."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "nfllib_amd", "csrc", "polymul4096_gfx950.s")
KNAME = "nflhip_polymul4096_asm"

# ------------------------------------------------------------------ register map
# SGPRs
S_KARG = "s[0:1]"
S_WGX, S_WGY = "s2", "s3"           # workgroup ids: x = poly index, y = modulus index
S_C, S_A, S_B, S_PSI, S_MC = "s[4:5]", "s[6:7]", "s[8:9]", "s[10:11]", "s[12:13]"
S_NM = "s14"
S_AROW, S_BROW, S_CROW, S_TW = "s[16:17]", "s[18:19]", "s[20:21]", "s[22:23]"
S_P, S_P2, S_P3 = "s[24:25]", "s[26:27]", "s[28:29]"
S_DELTA, S_MASK, S_C0 = "s30", "s31", "s15"      # delta, 0x3fffffff, 0xC0000000
S_MU2 = (32, 33)
S_NINV, S_NINVSH, S_W1N, S_W1NSH = (34, 35), (36, 37), (38, 39), (40, 41)
S_TMP = "s[42:43]"                   # scalar address scratch
S_CARRY = ["s[44:45]", "s[46:47]"]   # v_mad_u64_u32 carry-out, per stream
S_DUMMY = "s[48:49]"                 # dead carry-outs
S_BORROW = ["s[50:51]", "s[52:53]"]  # v_sub_co borrow, per stream
S_MCBUF = 56                         # s[56:83]: the ModConst record (28 dwords)
S_BASE2 = "s[84:85]"                 # scalar base of the current twiddle loads
S_R, S_BLK = "s88", "s89"            # r = logn - 12, blk = index of this 4096-word block inside its row
# K of each pass: twiddle index = (K << s) + (lane << s) + g (forward) / (K << s) - 1 - (lane << s) - g (inverse)
S_K = {"F1": "s90", "F2": "s91", "F3": "s92", "I1": "s93", "I2": "s94", "I3": "s95"}
NEXT_SGPR = 96

# VGPRs
V_TID = 0
V_OFF8 = 1        # tid*8 (global row offset)
V_L1W = 2         # LDS byte address, E1 write / E1' read : (t + (t>>4))*8
V_L1R = 3         # LDS byte address, E1 read / E2 write / E2' read / E1' write : (272*B + r)*8
V_L2R = 4         # LDS byte address, E2 read / E2' write : 17*t*8
V_BIDX = 5        # B = t >> 4
V_PHI = 6         # high dword of p (v_subb needs it in a VGPR)
V_A = 8           # v[8:39]    : a  (16 even-aligned pairs)
V_B = 40          # v[40:71]   : b
V_TW = 72         # v[72:131]  : 15 twiddle records (w lo, w hi, w' lo, w' hi)
V_T = [132, 150]  # per-stream temporaries (18 regs each)
NEXT_VGPR = 168   # 3 waves per SIMD
# address scratch lives in stream 1's temporaries (idle between butterflies)
V_TWO = V_T[1] + 1      # 32-bit per-lane twiddle offset
V_TWA = V_T[1] + 4      # 64-bit per-lane twiddle address
V_ZERO = V_T[0] + 15    # a persistent zero (the high half of stream 0's ZP pair)

LDS_BYTES = (4096 + 256) * 8


# Power / time ablations of the product kernels (tools/sessions/gpu_round3_g.sh; the results are WRONG by construction, the
# instruction stream is otherwise the shipped one): NFL_GEN_ABLATE = comma list of
#   tw0    every lane fetches the twiddle record of lane 0 (one cache line per wave instead of up to 64)
#   nolds  the exchanges through LDS are dropped (barriers stay)
#   row0   every workgroup works on one of the first 16 rows (operands and result stay in the L2)
#   nobar  the workgroup barriers are dropped as well
#   nobfly the butterflies of the register passes are dropped (memory, LDS and the point-wise step remain)
ABLATE = set(filter(None, os.environ.get("NFL_GEN_ABLATE", "").split(",")))
# scratchN (N a power of two): the n = 65536 pipeline's scratch rows a', b' of the WHOLE batch aliased onto N rows, i.e. the
# forward pass's writes and the block products' reads served by the on-die caches instead of HBM (round 5: what is the
# prize of a plan whose scratch never leaves the chip?)
SCRATCH_ALIAS = next((int(x[7:]) for x in ABLATE if x.startswith("scratch")), 0)
# bprimeN: the same question for rows of 32768 words (workload F): b' = NTT(b) makes a round trip through the context's
# scratch between the two launches of the composed product -- here over N row blocks instead of one per row
BPRIME_ALIAS = next((int(x[6:]) for x in ABLATE if x.startswith("bprime")), 0)
ALIAS_ROWS = ()   # set by build_row32k for its "_s" kinds: which of the row pointers s16 / s18 / s20 prologue16k aliases


SINGLE_STREAM = False   # ring mode: one butterfly at a time (18 temporaries instead of 36)
RING_RECOMPUTE_TWA = False


# In the ring-mode kernels (rows of 8192 / 16384 / 32768 words: one butterfly at a time, twiddle records streamed through
# the ring) the passes whose twiddle index depends on the THREAD (F3 / I1: global stages logn-4 .. logn-1, 15/16 of the
# table) read a lane-major copy of those stages: stage S = logn-4+s holds M 2^s records (M = n/16 = 256 << r), natural position
# (u << s) + g for thread-index u = 256 blk + t and group g, lane-major position g M + u.  A wave's 64 lanes then fetch 64
# CONSECUTIVE records per load (8 cache lines, all bytes used) instead of 64 records 16 << s bytes apart (up to 64 lines,
# 16 bytes used of each: 5.7 x the L2 -> L1 traffic over a pass).  The host lays the copy out (api.hip build_tables,
# DevTables::psi_lm); every other pass reads indices below n/16, which both layouts share.
#   ascending  (F3): index = K + ((256 c) << r) + t,          c  = 2^s - 1 + g            (K = 256 (2^r + blk))
#   descending (I1): index = K + ((256 c') << r) - 1 - t,     c' = 2^(s+1) - 2 - g        (K = (512 << r) - 256 blk)
# The lane part is the same for every stage: V_TWO = 16 t (ascending) or 16 (255 - t) (descending, base lowered by 256).
# Same-box A/B against the natural order (profiles/r03_lane_major_twiddles.txt): products +1 % (16384) / +3 % (8192) / +5 %
# (32768), inverse transforms +6 ... +16 %, forward +2 ... +8 %.  The 4096-word kernels (three workgroups per CU, all 15
# records of a pass resident) gain nothing from it (product +-0, pre-transformed product -2 %) and keep the natural table.
LANE_MAJOR = not os.environ.get("NFL_GEN_NATURAL_TWIDDLES")
SWAP_MAD = os.environ.get("NFL_GEN_SWAP_MAD", "")   # "" / "1" all (shipped), "0" none, "sgpr" / "vgpr": only the multiply-adds whose second factor is an SGPR / a VGPR
SWAP_MULHI = bool(os.environ.get("NFL_GEN_SWAP_MULHI"))
SPLIT32K = not os.environ.get("NFL_GEN_SERIAL_EXCHANGE")   # build_row32k: exchanges of one file under the arithmetic of the other


# (SGPRs of the second butterfly stream, idle in single-stream mode; s54/s55 are unused by the 4096-word map)
S_Q, S_SLAB = "s54", "s55"                  # sub-group index, byte offset of its LDS slab
S_K0 = {"F0": "s46", "I0": "s47"}
SLAB_BYTES = (4096 + 256) * 8


def configure(mode, groups=4):
    """Select the register map: "pair" = two interleaved butterflies, 15 twiddle records resident,
    168 VGPRs (3 waves/SIMD); "ring" = one butterfly at a time, 9-slot twiddle ring, 128 VGPRs (4 waves/SIMD)."""
    g = globals()
    if mode == "pair":
        g.update(SINGLE_STREAM=False, V_BIDX=5, V_PHI=6, V_A=8, V_B=40, V_TW=72, V_T=[132, 150], NEXT_VGPR=168,
                 NEXT_SGPR=96, LDS_BYTES=SLAB_BYTES, WG_SIZE=256)
        g.update(V_TWO=g["V_T"][1] + 1, V_TWA=g["V_T"][1] + 4, V_ZERO=g["V_T"][0] + 15)
    elif mode == "ringpair":
        # experiment (NFL_GEN_RINGPAIR=1): the 128-VGPR row kernels with TWO interleaved butterflies and a 5-slot ring instead
        # of one butterfly at a time and 9 slots; the twiddle address scratch lives in stream 1's temporaries
        g.update(SINGLE_STREAM=False, V_BIDX=5, V_PHI=6, V_A=8, V_B=40, V_TW=72, V_T=[92, 110], NEXT_VGPR=128, NEXT_SGPR=96,
                 RING_SLOTS=5, LDS_BYTES=groups * SLAB_BYTES, WG_SIZE=256 * groups, ROW_G=groups,
                 ROW_LG=groups.bit_length() - 1, RING_RECOMPUTE_TWA=True)
        g.update(V_TWO=g["V_T"][1] + 1, V_TWA=g["V_T"][1] + 4, V_ZERO=g["V_T"][0] + 15)
        g["S_K0"].update(F0="s98", I0="s99")   # (s46 / s47 are stream 1's carry pair here)
        g["NEXT_SGPR"] = 100
    else:
        g["S_K0"].update(F0="s46", I0="s47")
        g.update(SINGLE_STREAM=True, V_BIDX=5, V_PHI=6, V_TWO=7, V_TWA=8, V_A=10, V_B=42, V_TW=74, V_T=[110, 110],
                 NEXT_VGPR=128, NEXT_SGPR=96, RING_SLOTS=9, LDS_BYTES=groups * SLAB_BYTES, WG_SIZE=256 * groups,
                 ROW_G=groups, ROW_LG=groups.bit_length() - 1, RING_RECOMPUTE_TWA=False)
        g.update(V_ZERO=g["V_T"][0] + 15)


# ------------------------------------------------------------------ n = 65536: the three-role pipeline kernel
# Long rows need streaming radix-16 passes around the fused 4096-word block kernel, and the two kinds of work bound
# different resources (HBM vs integer VALU).  Kernels from different streams do not interleave on a CU in practice
# (DESIGN.md), so ONE launch carries all three kinds of workgroups, interleaved by workgroup index:
#   role 0  V   fused product of one 4096-word block of chunk j-1   (operands already passed through role 1/2)
#   role 1,2 F  forward radix-16 pass (global stages 0-3) of 256 columns of operand a / b of chunk j   (src -> dst)
#   role 3  I   inverse radix-16 pass (global stages 3-0, n^-1 folded in) of 256 columns of c of chunk j-2, in place
# Consecutive launches on one stream form the pipeline; inside a launch the roles are independent.
# kernarg: c_v a_v b_v psi mc | nm (logn unused) | cntV cntF cntI pad | fa_src fa_dst fb_src fb_dst inv_data pad
# grid: (28 * max(cnt), nm): wgx = 28*poly + w.
PIPE_LOGN = 16


# ------------------------------------------------------------------ one launch, rows pinned to an XCD
# The two-pass plan for rows that do not fit one CU moves every word 3 times (operand -> scratch -> scratch -> result):
# 9 word transfers per 3 algorithmic ones when the scratch lives in HBM.  Here the scratch of a row lives in the L2 of
# ONE XCD for the few microseconds between its producer and its consumer:
#   * row g of the batch (modulus-major: g = cm * batch + poly, so all XCDs work on the same modulus at the same time and
#     its twiddles stay in every L2) is job g / 8 of XCD g mod 8.  The grid is a fixed number of PERSISTENT workgroups;
#     each reads its XCC_ID once and then serves that XCD's jobs, whatever the placement of the workgroups.
#   * a job is 2 NSW forward streaming roles, then NV block products, then NSW inverse streaming roles.  Per XCD and kind
#     there is a CREDIT counter (roles that may start) and a TICKET counter (roles handed out, in job order).  A free
#     workgroup (its wave 0) reads the credits with one load, takes one with an atomic subtract (undone if it lost a race)
#     in the order inverse > product > forward -- inverse-first drains rows as fast as they mature -- and then draws the
#     next ticket of that kind.  Nothing spins on a shared word while work is available, and no atomic ever has to be
#     retried: hand-out is two fetch-and-adds.
#   * credits are posted by the role that completes a stage of a job (it sees the per-slot completion counter reach the
#     stage's size): forward -> NV product credits, product -> NSW inverse credits, inverse -> 2 NSW forward credits for
#     the job that reuses the scratch slot (R slots per XCD, job j uses slot j mod R).  Stages may complete out of job
#     order while tickets are in job order, so a role re-checks its own job's inputs before touching them; if k stages
#     have completed, the tickets of the first k jobs' roles of that stage have all been handed out (tickets are in
#     order), hence such a wait is only ever for roles that are already running: no deadlock.
#   * a role publishes "done" with one atomic add after all its stores were acknowledged by the L2 (s_waitcnt vmcnt(0) +
#     s_barrier); producer and consumer share the L2, nothing is written back in between.  The consumer's L1 is the one
#     cache that is not coherent with it, and it is kept out of the way by construction instead of by invalidation: every
#     row has its OWN scratch rows (the scratch mirrors the batch), so within a launch a scratch word is loaded by exactly
#     one workgroup after its last write, on a CU that either never touched the line or wrote it itself (block product:
#     reads a'[k], writes c'[k] over it -- write-through keeps its own L1 current); L1s start a launch invalidated.
#     What was measured on the way (tools/probes/l2_flag_probe.hip, profiles/README): workgroup-scope (sc0) loads hit
#     in the L1 and never see another CU's update; device-scope (sc1) loads and atomics are served memory-side (0.15 -
#     0.5 us) -- scratch read with sc1 loads was correct but moved MORE HBM bytes than the chunked pipeline (3.8x vs 3.3x
#     the algorithmic bytes); `buffer_inv sc0` does not reliably drop stale lines (wrong words in 4 of 9 runs).
#     The ring only bounds the rows in flight (R per domain): its slots index the completion counters.
# Kernel arguments after the standard seven: rows, batch, ceil(2^32 / batch), log2 D | Rlog, -, spin limit, - | scrA, scrB,
# ctl, trace buffer (or null).  D = scheduling domains per XCD (each with its own record, jobs and ring; they only share the
# L2).  ctl: +64 + 4 xcd: workgroups that joined; the record of domain d = xcd + 8 sub at byte 4096 + 69632 d (zeroed by the host):
#   +0 credits {forward (biased by the initial min(R, jobs) * 2 NSW), product, inverse}, +12 exit flag, +16 trace count
#   +128 tickets {forward, product, inverse}      +256 + 16 slot: completed {forward, product, inverse} roles (all epochs)
FUSED_NT = int(os.environ.get("NFL_FUSED_NT", "1"))
FUSED_LIFO = False                       # scratch rows come from a per-XCD pool, lowest free slot first (lifo_* below)
FUSED_LOADS = ""                         # modifier of the scratch loads: " sc1" = device scope (L1 bypass), "" = plain after a buffer_inv sc0
LDS_TICKET = (4096 + 256) * 8           # 64 B behind the exchange slab: wave 0's decision and the running role's completion record
S_FMT, S_F = "s4", "s5"
S_X2ROW, S_K0ROW, S_K1ROW, S_O1ROW = "s[54:55]", "s[96:97]", "s[98:99]", "s[100:101]"


# ------------------------------------------------------------------ transform-fused pipelines, rows of 8192 / 16384 words
# The same four pipelines on the row-resident register map of build_row16k (ring mode: 128 VGPRs, one butterfly at a time,
# twiddle records streaming through the 9-slot ring; ROW_G sub-groups of 256 threads, one outer radix-ROW_G pass F0 / I0
# around the 4096-word passes).  One workgroup per (batch element, modulus) row of exactly 4096 ROW_G words.  The ring is
# empty between a transform and the next one, so the 32 registers of a key row's 16 words live in ITS slots: the key is
# loaded behind the last forward record, the multiply-add lands in the key's registers (x' stays for the second result), and
# the next transform's ring is primed once the result's stores have been issued.  The exchanges are the plain ones of
# build_row16k (write, barrier, read): the split-phase schedules of the product kernels are tied to their two-operand shape.
# kernarg as ARGS_FUSED; both grids of prologue_fused.
S_X2ROW16 = "s[52:53]"     # (stream 1's borrow pair: idle in single-stream mode)


def set(**values):
    """rebind configuration names between kernel families (main.py): cfg.set(NEXT_SGPR=102)"""
    globals().update(values)
