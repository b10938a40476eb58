"""CPU restatement of the reference's random constructors as pure functions of the random words they consume.

TEST INFRASTRUCTURE ONLY (see oracle/nfl_oracle.h): nothing under nfllib_amd/ or include/ may import this.

The reference draws its randomness from a process-global Salsa20 stream (lib/prng/fastrandombytes.cpp:17-37), so
its samplers cannot be replayed directly; what CAN be pinned is the map from the consumed random words to the
coefficients.  tools/gen_golden.py records, from the real reference, pairs (the bytes the next fastrandombytes call
returns, the polynomial the constructor then builds from exactly those bytes) -- see oracle/ref_shim.cpp
`nflref_sample_replay` -- and tests/test_oracle_golden.py checks these functions against them.  The device kernels
(nfllib_amd/csrc/kernels_sample.hip) implement the same maps over a ChaCha20 stream, restated here as well.

Parity status: uniform / non_uniform / ZO_dist rules PINNED against the reference (tests/golden/samplers.npz);
hwt_dist and gaussian are distribution-level restatements (the reference's own procedures are not functions of a
replayable word sequence): their tests are statistical and compare against samples of the real reference.
"""
import numpy as np

_U64 = np.uint64


def _mask_bits(x):
    """(1 << (floor(log2 x) + 1)) - 1 for x >= 1, as core.hpp:165-166 / 223-224 compute it"""
    return (1 << int(x).bit_length()) - 1


def uniform(words, P):
    """poly::set(uniform) core.hpp:152-188.  words: [..., nm, n] random words of the limb width; P: the moduli."""
    words = np.asarray(words)
    out = np.empty_like(words)
    for cm, p in enumerate(P):
        mask = _mask_bits(p)
        v = words[..., cm, :].astype(object) & mask
        v = np.where(v >= p, v - p, v)
        out[..., cm, :] = v.astype(words.dtype)
    return out


def non_uniform(words, P, upper_bound, amplifier=1, dtype=np.uint64):
    """poly::set(non_uniform) core.hpp:195-277.  words: [..., n] one word per coefficient; result [..., nm, n]."""
    w = np.asarray(words).astype(object)
    t = 2 * upper_bound - 1
    tmp = w & _mask_bits(t)
    tmp = np.where(tmp >= t, tmp - t, tmp)
    neg = tmp >= upper_bound
    out = np.empty(w.shape[:-1] + (len(P), w.shape[-1]), dtype=object)
    for cm, p in enumerate(P):
        # core.hpp:243-244 / 267-268: p + tmp*amp - (2ub-1)*amp for the negatives, tmp*amp otherwise
        out[..., cm, :] = np.where(neg, p - (t - tmp) * amplifier, tmp * amplifier)
    return out.astype(dtype)


def zo_dist(rnd_bytes, P, rho, canonical=True, dtype=np.uint64):
    """poly::set(ZO_dist) core.hpp:330-340: byte <= rho ? (p-1) + (byte & 2) : 0.  The reference stores +1 as p+1
    (canonical=False reproduces that); the engine stores the canonical 1."""
    b = np.asarray(rnd_bytes).astype(object)
    out = np.empty(b.shape[:-1] + (len(P), b.shape[-1]), dtype=object)
    for cm, p in enumerate(P):
        v = np.where(b <= rho, (p - 1) + (b & 2), 0)
        if canonical:
            v = np.where(v >= p, v - p, v)
        out[..., cm, :] = v
    return out.astype(dtype)


def centered(data, P):
    """signed representatives in (-p/2, p/2] of every residue row: [..., nm, n] -> int64 (small values only)"""
    d = np.asarray(data).astype(object)
    out = np.empty(d.shape, dtype=object)
    for cm, p in enumerate(P):
        v = d[..., cm, :]
        out[..., cm, :] = np.where(v > p // 2, v - p, v)
    return out.astype(np.int64)


# ---- the device's keystream: ChaCha20 (djb layout: 64-bit counter, 64-bit nonce), kernels_sample.hip -------------
SECONDARY_COUNTER = 1 << 63   # first block counter of the secondary stream (kernels_sample.hip, lazy-precision Gaussian)
# domain separation (kernels_sample.hip, ChaChaKey::dom): bits 56..62 of the block counter carry the distribution's tag,
# so calls of different distributions that share (key, stream_id) never read the same keystream word
DOMAIN = {"raw": 0, "uniform": 1, "bounded": 2, "zo": 3, "hwt": 4, "gauss": 5,
          # the narrow draws (kernels_sample.hip "NARROW DRAWS"): a value reads a LANE of the keystream, in a domain of its own
          "uniform_narrow": 6, "gauss32": 7, "gauss32_ref": 8}


def domain_base(name):
    return DOMAIN[name] << 56


def chacha20_words(key32, stream_id, first_word, nwords, counter_base=0):
    """64-bit little-endian words [first_word, first_word + nwords) of the keystream (key32, nonce = stream_id);
    counter_base is added to every block counter (the secondary stream starts at block 2^63)."""
    key = np.frombuffer(bytes(key32), dtype="<u4").astype(np.uint32)
    assert key.size == 8
    fb, lb = first_word // 8, (first_word + nwords + 7) // 8
    ctr = np.arange(fb, lb, dtype=np.uint64) + np.uint64(counter_base)
    nb = ctr.size
    s = np.empty((16, nb), dtype=np.uint32)
    s[0], s[1], s[2], s[3] = 0x61707865, 0x3320646E, 0x79622D32, 0x6B206574
    for i in range(8):
        s[4 + i] = key[i]
    s[12] = (ctr & _U64(0xFFFFFFFF)).astype(np.uint32)
    s[13] = (ctr >> _U64(32)).astype(np.uint32)
    s[14] = np.uint32(stream_id & 0xFFFFFFFF)
    s[15] = np.uint32((stream_id >> 32) & 0xFFFFFFFF)
    x = s.copy()

    def rotl(v, r):
        return (v << np.uint32(r)) | (v >> np.uint32(32 - r))

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16)
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12)
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8)
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7)

    with np.errstate(over="ignore"):
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += s
    lo, hi = x[0::2].astype(np.uint64), x[1::2].astype(np.uint64)        # [8, nb]
    words = (lo | (hi << _U64(32))).T.reshape(-1)                          # block-major
    off = first_word - fb * 8
    return words[off:off + nwords].copy()


def chacha20_lanes(key32, stream_id, first_lane, nlanes, lane_bytes, counter_base=0):
    """lanes [first_lane, first_lane + nlanes) of the keystream read as little-endian integers of lane_bytes bytes each
    (lane g = bytes [g * lane_bytes, (g + 1) * lane_bytes) of the stream): what the narrow draws consume"""
    per = 8 // lane_bytes
    fw, lw = first_lane // per, (first_lane + nlanes + per - 1) // per
    words = chacha20_words(key32, stream_id, fw, lw - fw, counter_base=counter_base)
    lanes = words.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[lane_bytes])   # (little-endian host)
    off = first_lane - fw * per
    return lanes[off:off + nlanes].copy()


def uniform_narrow_words(key32, stream_id, first_word, nwords, limb_bits):
    """the random lanes poly(uniform) consumes under NFLHIP_DIST_NARROW: residue word g reads the limb-width lane g"""
    return chacha20_lanes(key32, stream_id, first_word, nwords, limb_bits // 8, counter_base=domain_base("uniform_narrow"))


def gaussian_pmf(sigma, center, x_min, entries):
    """P(X = x) of the tail-cut discrete Gaussian on [x_min, x_min + entries) (FastGaussianNoise.hpp:296-330)"""
    x = np.arange(x_min, x_min + entries, dtype=np.float64)
    rho = np.exp(-((x - center) ** 2) / (2.0 * sigma * sigma))
    return rho / rho.sum()


def gaussian_words(key32, stream_id, first_coef, ncoef, W):
    """The W-word uniform number of every coefficient g in [first_coef, first_coef + ncoef), most significant word first,
    as the device defines it: word 0 = primary stream word g; word k >= 1 = secondary stream word (W-1)*g + k-1 (the
    device only ever reads those when word 0 ties with a table entry)."""
    r = np.empty((ncoef, W), dtype=np.uint64)
    r[:, 0] = chacha20_words(key32, stream_id, first_coef, ncoef, counter_base=domain_base("gauss"))
    if W > 1:
        r[:, 1:] = chacha20_words(key32, stream_id, (W - 1) * first_coef, (W - 1) * ncoef,
                                  counter_base=SECONDARY_COUNTER | domain_base("gauss")).reshape(ncoef, W - 1)
    return r


def gaussian_words_narrow(key32, stream_id, first_coef, ncoef, W):
    """The W-word uniform number of coefficient g under the 32-bit draw (nflhip_gauss_set_draw_bits(g, 32)): word 0 =
    (32-bit lane g of stream "gauss32") << 32 | 32-bit lane g of stream "gauss32_ref" (the device reads the second lane only
    when the first ties with the top half of a table entry); words k >= 1 as under the 64-bit draw, from the secondary stream of
    the "gauss32" domain."""
    r = np.empty((ncoef, W), dtype=np.uint64)
    hi = chacha20_lanes(key32, stream_id, first_coef, ncoef, 4, counter_base=domain_base("gauss32")).astype(np.uint64)
    lo = chacha20_lanes(key32, stream_id, first_coef, ncoef, 4, counter_base=domain_base("gauss32_ref")).astype(np.uint64)
    r[:, 0] = (hi << _U64(32)) | lo
    if W > 1:
        r[:, 1:] = chacha20_words(key32, stream_id, (W - 1) * first_coef, (W - 1) * ncoef,
                                  counter_base=SECONDARY_COUNTER | domain_base("gauss32")).reshape(ncoef, W - 1)
    return r


def gaussian_from_table(r_words, table, x_min):
    """inversion through a cumulative table of multi-word entries (most significant word first), as the device does:
    x = x_min + #{k : table[k] <= r}.  r_words: [..., W] uint64, table: [entries, W] uint64."""
    W = table.shape[1]
    tab = [int.from_bytes(b"".join(int(v).to_bytes(8, "big") for v in row), "big") for row in table]
    r = np.asarray(r_words).reshape(-1, W)
    out = np.empty(r.shape[0], dtype=np.int64)
    import bisect
    for i, row in enumerate(r):
        val = int.from_bytes(b"".join(int(v).to_bytes(8, "big") for v in row), "big")
        out[i] = x_min + bisect.bisect_right(tab, val)
    return out.reshape(np.asarray(r_words).shape[:-1])


def gaussian_reference_decode(raw_calls, call_words, barriers, wp, rlen, x0):
    """What FastGaussianNoise<uint8_t, ., 2>::getNoise (FastGaussianNoise.hpp:477-595) makes of its uniform bytes, stated
    as inversion: a sample is x0 + #{barriers <= noise}; the reference looks at one byte, at two if some barrier starts
    with that byte, and at the full wp-byte number only if some barrier starts with those two, and it starts a fresh
    buffer (the next fastrandombytes call) as soon as fewer than wp bytes + 1 are left of the current one.
    raw_calls: [ncalls, call_bytes] uint8; barriers: ascending list of wp-byte big-endian bytes objects.
    Returns (samples, prefixes): the values and the bytes each one consumed."""
    import bisect
    first = {b[0] for b in barriers}
    first2 = {(b[0], b[1]) for b in barriers}
    out, prefixes = [], []
    call, pos, used = 0, 0, 0
    while len(out) < rlen:
        buf = raw_calls[call]
        b1 = int(buf[pos])
        if b1 not in first:
            k = 1
        elif (b1, int(buf[pos + 1])) not in first2:
            k = 2
        else:
            k = wp
        pre = bytes(buf[pos:pos + k])
        out.append(x0 + bisect.bisect_right(barriers, pre + b"\xff" * (wp - k)))
        prefixes.append(pre)
        pos += k
        used += k
        if used + wp >= call_words:
            call, pos, used = call + 1, 0, 0
    return np.array(out, dtype=np.int64), prefixes
