"""TEST INFRASTRUCTURE: a small interpreter for the subset of gfx950 assembly that the row-kernel generators
(tools/gen_row1024_u32_asm.py, gen_row128_u16_asm.py, gen_row8_u32_asm.py) emit, so that the GENERATED kernels can be
checked against the oracle on a machine without a GPU (tests/test_asm_emulated.py, `-m "not gpu"`).

It executes the text the generators produce: one numpy vector of 64 lanes per VGPR, scalar registers, VCC / EXEC / SCC,
a flat "device memory" made of registered buffers, one LDS array per workgroup, waves of a workgroup interleaved at
s_barrier.  Timing, hazards and wait counts are ignored (every memory operation completes at once); an instruction the
interpreter does not know raises, so a generator change that needs more of the ISA fails loudly here.
"""
import re

import numpy as np

M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1


class Memory:
    """registered buffers at fake device addresses"""

    def __init__(self):
        self.bufs = []
        self.next = 0x100000000

    def add(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        base = self.next
        self.bufs.append((base, a))
        self.next += (a.size + 0xFFFF) & ~0xFFFF
        self.next += 0x10000
        return base

    def find(self, addr, n):
        for base, a in self.bufs:
            if base <= addr and addr + n <= base + a.size:
                return a, addr - base
        raise RuntimeError("emulated access outside every buffer: 0x%x (+%d)" % (addr, n))

    def read(self, addr, n):
        a, o = self.find(addr, n)
        return bytes(a[o:o + n])

    def write(self, addr, data):
        a, o = self.find(addr, len(data))
        a[o:o + len(data)] = np.frombuffer(data, dtype=np.uint8)


def parse_program(text):
    """-> (instructions, labels): instructions = list of (mnemonic, operands, modifiers); reads up to .Lfunc_end"""
    ins, labels = [], {}
    for raw in text.split("\n"):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.startswith(".Lfunc_end"):
            break
        if line.endswith(":"):
            labels[line[:-1]] = len(ins)
            continue
        if line.startswith("."):
            continue
        m = re.match(r"([a-z_0-9]+)\s*(.*)", line)
        mn, rest = m.group(1), m.group(2)
        mods = {}
        pat = r"\b(offset|dst_sel|dst_unused|src0_sel|src1_sel):(\S+)"
        for key, val in re.findall(pat, rest):
            mods[key] = val
        rest = re.sub(pat, "", rest)
        for flag in ("sc0", "sc1", "nt", "glc"):
            if re.search(r"\b%s\b" % flag, rest):
                mods[flag] = True
                rest = re.sub(r"\b%s\b" % flag, "", rest)
        if mn == "s_waitcnt":
            ops = []
        else:
            ops = [o.strip() for o in rest.split(",") if o.strip()]
        ins.append((mn, ops, mods))
    return ins, labels


class Wave:
    def __init__(self, prog, labels, mem, lds, kernarg_addr, wg_id, wave_in_wg, wg_y=0):
        self.prog, self.labels, self.mem, self.lds = prog, labels, mem, lds
        self.v = np.zeros((256, 64), dtype=np.uint64)     # (kept as uint64, masked to 32 bits)
        self.s = [0] * 108
        self.vcc = 0
        self.exec = M64
        self.scc = 0
        self.pc = 0
        self.s[0], self.s[1] = kernarg_addr & M32, kernarg_addr >> 32
        self.s[2] = wg_id
        self.s[3] = wg_y
        self.v[0] = np.arange(64, dtype=np.uint64) + 64 * wave_in_wg
        self.done = False

    # ---- operands
    def lanes(self):
        return self.mask_of(self.exec)

    @staticmethod
    def mask_of(bits):
        return np.unpackbits(np.array([bits], dtype="<u8").view(np.uint8), bitorder="little").astype(bool)

    @staticmethod
    def bits_of(flags):
        return int(np.packbits(flags, bitorder="little").view("<u8")[0])

    def rd32(self, op):
        """32-bit source operand -> numpy uint64 vector (or scalar broadcast)"""
        if re.fullmatch(r"v\d+", op):
            return self.v[int(op[1:])].copy()
        if re.fullmatch(r"s\d+", op):
            return np.full(64, self.s[int(op[1:])], dtype=np.uint64)
        if op == "vcc_lo":
            return np.full(64, self.vcc & M32, dtype=np.uint64)
        return np.full(64, int(op, 0) & M32, dtype=np.uint64)

    def rd64(self, op):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        if m:
            lo = int(m.group(1))
            return self.v[lo] | (self.v[lo + 1] << np.uint64(32))
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
        if m:
            lo = int(m.group(1))
            return np.full(64, self.s[lo] | (self.s[lo + 1] << 32), dtype=np.uint64)
        return np.full(64, int(op, 0) & M64, dtype=np.uint64)

    def srd(self, op):
        if re.fullmatch(r"s\d+", op):
            return self.s[int(op[1:])]
        return int(op, 0) & M32

    def srd64(self, op):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
        if m:
            lo = int(m.group(1))
            return self.s[lo] | (self.s[lo + 1] << 32)
        if op == "vcc":
            return self.vcc
        if op == "exec":
            return self.exec
        return int(op, 0) & M64

    def swr(self, op, val):
        self.s[int(op[1:])] = val & M32

    def swr64(self, op, val):
        if op == "vcc":
            self.vcc = val & M64
        elif op == "exec":
            self.exec = val & M64
        else:
            lo = int(re.fullmatch(r"s\[(\d+):(\d+)\]", op).group(1))
            self.s[lo], self.s[lo + 1] = val & M32, (val >> 32) & M32

    def wr32(self, op, val):
        r = int(op[1:])
        act = self.lanes()
        self.v[r][act] = (val & np.uint64(M32))[act]

    def wr64(self, op, val):
        lo = int(re.fullmatch(r"v\[(\d+):(\d+)\]", op).group(1))
        act = self.lanes()
        self.v[lo][act] = (val & np.uint64(M32))[act]
        self.v[lo + 1][act] = (val >> np.uint64(32))[act]

    @staticmethod
    def sel(vec, how):
        if how in (None, "DWORD"):
            return vec
        if how == "WORD_0":
            return vec & np.uint64(0xFFFF)
        if how == "WORD_1":
            return (vec >> np.uint64(16)) & np.uint64(0xFFFF)
        raise RuntimeError("SDWA select %s" % how)

    def gaddr(self, vaddr, saddr, mods):
        """global_* addressing: (scalar base + immediate, per-lane offsets); `off` = a 64-bit address in a VGPR pair"""
        imm = int(mods.get("offset", "0"), 0)
        if saddr == "off":
            return imm, self.rd64(vaddr)
        return self.srd64(saddr) + imm, self.v[int(vaddr[1:])]

    # ---- execution
    def run(self):
        """generator: yields at every s_barrier, returns at s_endpgm"""
        prog = self.prog
        u = np.uint64
        while True:
            mn, ops, mods = prog[self.pc]
            self.pc += 1
            if mn in ("s_waitcnt", "s_nop", "s_sleep"):
                continue
            if mn == "s_endpgm":
                self.done = True
                return
            if mn == "s_barrier":
                yield
                continue
            if mn.startswith("s_load_dword"):
                n = 1 if mn == "s_load_dword" else int(mn.split("x")[1])
                addr = self.srd64(ops[1]) + int(ops[2], 0)
                data = np.frombuffer(self.mem.read(addr, 4 * n), dtype=np.uint32)
                lo = int(re.match(r"s\[?(\d+)", ops[0]).group(1))
                for k in range(n):
                    self.s[lo + k] = int(data[k])
                continue
            if mn in ("s_mov_b32",):
                self.swr(ops[0], self.srd(ops[1])); continue
            if mn == "s_mov_b64":
                self.swr64(ops[0], self.srd64(ops[1])); continue
            if mn in ("s_add_u32", "s_sub_u32", "s_addc_u32", "s_subb_u32"):
                a, b = self.srd(ops[1]), self.srd(ops[2])
                if mn == "s_add_u32":
                    r = a + b; self.scc = r >> 32
                elif mn == "s_addc_u32":
                    r = a + b + self.scc; self.scc = r >> 32
                elif mn == "s_sub_u32":
                    r = a - b; self.scc = 1 if b > a else 0
                else:
                    r = a - b - self.scc; self.scc = 1 if b + self.scc > a else 0
                self.swr(ops[0], r); continue
            if mn in ("s_lshl_b32", "s_lshr_b32", "s_mul_i32", "s_mul_hi_u32", "s_and_b32", "s_or_b32", "s_min_u32"):
                a, b = self.srd(ops[1]), self.srd(ops[2])
                r = {"s_lshl_b32": lambda: a << (b & 31), "s_lshr_b32": lambda: a >> (b & 31), "s_mul_i32": lambda: a * b,
                     "s_mul_hi_u32": lambda: (a * b) >> 32, "s_and_b32": lambda: a & b, "s_or_b32": lambda: a | b,
                     "s_min_u32": lambda: min(a, b)}[mn]()
                if mn in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32"):
                    self.scc = 1 if (r & M32) else 0
                self.swr(ops[0], r); continue
            if mn.startswith("s_cmp_"):
                a, b = self.srd(ops[0]), self.srd(ops[1])
                self.scc = int({"s_cmp_lt_u32": a < b, "s_cmp_eq_u32": a == b, "s_cmp_lg_u32": a != b, "s_cmp_ge_u32": a >= b,
                                "s_cmp_gt_u32": a > b, "s_cmp_le_u32": a <= b}[mn]); continue
            if mn in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch", "s_cbranch_execz"):
                take = {"s_cbranch_scc1": self.scc == 1, "s_cbranch_scc0": self.scc == 0, "s_branch": True,
                        "s_cbranch_execz": self.exec == 0}[mn]
                if take:
                    self.pc = self.labels[ops[0]]
                continue
            if mn == "s_cselect_b32":
                self.swr(ops[0], self.srd(ops[1]) if self.scc else self.srd(ops[2])); continue
            if mn == "s_cselect_b64":
                self.swr64(ops[0], self.srd64(ops[1]) if self.scc else self.srd64(ops[2])); continue
            if mn == "s_and_saveexec_b64":
                old = self.exec
                self.exec &= self.srd64(ops[1])
                self.swr64(ops[0], old)
                self.scc = 1 if self.exec else 0
                continue
            # ---- vector ALU
            if mn == "v_readfirstlane_b32":
                first = next(i for i in range(64) if (self.exec >> i) & 1) if self.exec else 0
                self.swr(ops[0], int(self.v[int(ops[1][1:])][first])); continue
            if mn in ("v_mov_b32_e32",):
                self.wr32(ops[0], self.rd32(ops[1])); continue
            if mn == "v_mad_u64_u32":
                prod = self.rd32(ops[2]) * self.rd32(ops[3])
                r = prod + self.rd64(ops[4])           # (wraps at 64 bits)
                act = self.lanes()
                self.swr64(ops[1], self.bits_of(act & (r < prod)))
                self.wr64(ops[0], r); continue
            if mn in ("v_mul_u32_u24_sdwa", "v_mul_u32_u24_e32"):
                a = self.sel(self.rd32(ops[1]), mods.get("src0_sel")) & u(0xFFFFFF)
                b = self.sel(self.rd32(ops[2]), mods.get("src1_sel")) & u(0xFFFFFF)
                self.wr32(ops[0], a * b); continue
            if mn == "v_mad_u32_u24":
                self.wr32(ops[0], (self.rd32(ops[1]) & u(0xFFFFFF)) * (self.rd32(ops[2]) & u(0xFFFFFF)) + self.rd32(ops[3])); continue
            if mn == "v_lshl_add_u32":
                self.wr32(ops[0], (self.rd32(ops[1]) << (self.rd32(ops[2]) & u(31))) + self.rd32(ops[3])); continue
            if mn == "v_alignbit_b32":
                w = (self.rd32(ops[1]) << u(32)) | self.rd32(ops[2])
                self.wr32(ops[0], w >> (self.rd32(ops[3]) & u(31))); continue
            if mn == "v_mul_hi_u32":
                self.wr32(ops[0], (self.rd32(ops[1]) * self.rd32(ops[2])) >> u(32)); continue
            if mn == "v_mul_lo_u32":
                self.wr32(ops[0], self.rd32(ops[1]) * self.rd32(ops[2])); continue
            if mn.startswith("v_cmp_") and mn.endswith("_u32_e32"):
                a, b = self.rd32(ops[1]), self.rd32(ops[2])
                r = {"gt": a > b, "lt": a < b, "ge": a >= b, "le": a <= b, "eq": a == b, "ne": a != b}[mn.split("_")[2]]
                act = self.lanes()
                self.vcc = self.bits_of(act & r); continue
            if mn in ("v_cndmask_b32_e64", "v_cndmask_b32_e32"):
                m = self.mask_of(self.srd64(ops[3]) if len(ops) > 3 else self.vcc)
                self.wr32(ops[0], np.where(m, self.rd32(ops[2]), self.rd32(ops[1]))); continue
            if mn in ("v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_min_u32_e32", "v_and_b32_e32", "v_or_b32_e32",
                      "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_max_u32_e32", "v_xor_b32_e32"):
                a, b = self.rd32(ops[1]), self.rd32(ops[2])
                r = {"v_add_u32_e32": lambda: a + b, "v_sub_u32_e32": lambda: a - b, "v_subrev_u32_e32": lambda: b - a,
                     "v_min_u32_e32": lambda: np.minimum(a, b), "v_max_u32_e32": lambda: np.maximum(a, b),
                     "v_and_b32_e32": lambda: a & b, "v_or_b32_e32": lambda: a | b, "v_xor_b32_e32": lambda: a ^ b,
                     "v_lshlrev_b32_e32": lambda: b << (a & u(31)), "v_lshrrev_b32_e32": lambda: b >> (a & u(31))}[mn]()
                self.wr32(ops[0], r & u(M32)); continue
            if mn == "v_lshl_add_u64":
                self.wr64(ops[0], ((self.rd64(ops[1]) << (self.rd32(ops[2]) & u(63))) + self.rd64(ops[3])) & u(M64)); continue
            if mn in ("v_add_co_u32_e64", "v_add_co_u32_e32", "v_addc_co_u32_e64", "v_addc_co_u32_e32", "v_sub_co_u32_e64",
                      "v_sub_co_u32_e32", "v_subb_co_u32_e64", "v_subb_co_u32_e32", "v_subrev_co_u32_e32", "v_subbrev_co_u32_e32"):
                a, b = self.rd32(ops[2]), self.rd32(ops[3])
                cin = self.mask_of(self.srd64(ops[4])).astype(np.uint64) if len(ops) > 4 else u(0)
                if "rev" in mn:
                    a, b = b, a
                if mn.startswith("v_add"):
                    r = a + b + cin
                    cout = r > u(M32)
                else:
                    r = a - b - cin            # (wraps in 64 bits; the low word is what counts)
                    cout = (b + cin) > a
                act = self.lanes()
                self.swr64(ops[1], self.bits_of(act & cout))
                self.wr32(ops[0], r & u(M32)); continue
            # ---- memory
            if mn.startswith("global_load_"):
                n = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "ushort": 2, "ubyte": 1}[mn[len("global_load_"):]]
                base, off = self.gaddr(ops[1], ops[2], mods)
                lo = int(re.match(r"v\[?(\d+)", ops[0]).group(1))
                act = self.lanes()
                for lane in range(64):
                    if not act[lane]:
                        continue
                    raw = self.mem.read(base + int(off[lane]), n)
                    if n >= 4:
                        for k, w in enumerate(np.frombuffer(raw, dtype=np.uint32)):
                            self.v[lo + k][lane] = int(w)
                    else:
                        self.v[lo][lane] = int.from_bytes(raw, "little")
                continue
            if mn.startswith("global_store_"):
                n = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "short": 2, "byte": 1}[mn[len("global_store_"):]]
                base, off = self.gaddr(ops[0], ops[2], mods)
                lo = int(re.match(r"v\[?(\d+)", ops[1]).group(1))
                act = self.lanes()
                for lane in range(64):
                    if not act[lane]:
                        continue
                    if n >= 4:
                        data = b"".join(int(self.v[lo + k][lane]).to_bytes(4, "little") for k in range(n // 4))
                    else:
                        data = (int(self.v[lo][lane]) & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
                    self.mem.write(base + int(off[lane]), data)
                continue
            if mn in ("ds_write_b32", "ds_read_b32"):
                imm = int(mods.get("offset", "0"), 0)
                act = self.lanes()
                if mn == "ds_write_b32":
                    addr, data = self.v[int(ops[0][1:])], self.v[int(ops[1][1:])]
                    for lane in range(64):
                        if act[lane]:
                            self.lds[(int(addr[lane]) + imm) // 4] = int(data[lane])
                else:
                    addr = self.v[int(ops[1][1:])]
                    r = int(ops[0][1:])
                    for lane in range(64):
                        if act[lane]:
                            self.v[r][lane] = self.lds[(int(addr[lane]) + imm) // 4]
                continue
            if mn in ("ds_write_b64", "ds_read_b64"):
                imm = int(mods.get("offset", "0"), 0)
                act = self.lanes()
                if mn == "ds_write_b64":
                    addr = self.v[int(ops[0][1:])]
                    lo = int(re.match(r"v\[(\d+)", ops[1]).group(1))
                    for lane in range(64):
                        if act[lane]:
                            w = (int(addr[lane]) + imm) // 4
                            self.lds[w], self.lds[w + 1] = int(self.v[lo][lane]), int(self.v[lo + 1][lane])
                else:
                    addr = self.v[int(ops[1][1:])]
                    lo = int(re.match(r"v\[(\d+)", ops[0]).group(1))
                    for lane in range(64):
                        if act[lane]:
                            w = (int(addr[lane]) + imm) // 4
                            self.v[lo][lane], self.v[lo + 1][lane] = self.lds[w], self.lds[w + 1]
                continue
            raise RuntimeError("emulator: unknown instruction %s %s" % (mn, ops))


def run_kernel(asm_text, mem, kernarg, grid, lds_bytes, waves_per_wg=4):
    """execute the workgroups of the kernel in asm_text one after the other; grid = gx or (gx, gy); kernarg = bytes"""
    prog, labels = parse_program(asm_text)
    karg = mem.add(np.frombuffer(kernarg + b"\0" * 64, dtype=np.uint8).copy())
    gx, gy = grid if isinstance(grid, tuple) else (grid, 1)
    for wg in range(gx * gy):
        lds = np.zeros(lds_bytes // 4 + 16, dtype=np.uint32)
        waves = [Wave(prog, labels, mem, lds, karg, wg % gx, w, wg // gx) for w in range(waves_per_wg)]
        gens = [w.run() for w in waves]
        live = list(range(waves_per_wg))
        while live:
            for i in list(live):
                try:
                    next(gens[i])        # runs to the next barrier
                except StopIteration:
                    live.remove(i)


def device_tables(limb_bits, n, nm, prm):
    """the twiddle table (Tw<T>: {psi^bitrev(k), Shoup companion}) and the ModConst<T> records exactly as
    nfllib_amd/csrc/api.hip build_tables lays them out on the device (negacyclic case), from params<T>"""
    wb, logn = limb_bits, n.bit_length() - 1
    dt = prm.dtype
    psi = np.zeros((nm, n, 2), dtype=dt)
    mc = np.zeros((nm, 14), dtype=dt)
    for cm in range(nm):
        p = int(prm.P[cm])
        phi = int(prm.primitive_roots[cm])
        for _ in range(prm.kmax_log2 - logn):
            phi = phi * phi % p
        assert pow(phi, n, p) == p - 1
        for k in range(n):
            e = int(format(k, "0%db" % logn)[::-1], 2) if logn else 0
            w = pow(phi, e, p)
            psi[cm, k] = (w, (w << wb) // p)
        ninv = int(prm.invkmax[cm]) * (prm.kmax // n) % p
        assert ninv * n % p == 1
        w1n = int(psi[cm, 1, 0]) * ninv % p
        beta = (1 << 64) % p
        bits = p.bit_length()
        rec = [p, 2 * p, (1 << (2 * wb - 4)) // p, ninv, (ninv << wb) // p, w1n, (w1n << wb) // p, beta, (beta << wb) // p,
               0, 0, (1 << bits) - 1, (1 << (wb - 2)) - p, (1 << (2 * wb - 3)) // p]   # (yinv: CRT only, unused by the row kernels)
        mc[cm] = [r & ((1 << wb) - 1) for r in rec]
    return psi, mc


def run_row_kernel(asm_path, limb_bits, n, nm, prm, a, b, rows_per_wg, with_magic):
    """a, b: (batch, nm, n) arrays -> the kernel's output array (same shape)"""
    import struct
    mem = Memory()
    psi, mc = device_tables(limb_bits, n, nm, prm)
    c = np.zeros_like(a)
    pa, pb, pc, ppsi, pmc = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(psi), mem.add(mc)
    rows = a.shape[0] * nm
    magic = 0 if (nm == 1 or not with_magic) else ((1 << 32) // nm + 1)
    kernarg = struct.pack("<5Q2IQ", pc, pa, pb, ppsi, pmc, nm, magic, rows)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    run_kernel(text, mem, kernarg, (rows + rows_per_wg - 1) // rows_per_wg, lds)
    out, _ = mem.find(pc, c.nbytes)
    return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()


def run_block_kernel(asm_path, n, nm, prm, a, b, block_log, count=None):
    """the 64-bit block kernels of tools/gen_polymul_asm.py (kernarg: dst, a, b, psi, mc, nm, logn[, count]; grid =
    (blocks of the batch, nm); 2^block_log words per workgroup, 16 per thread).  count (the two-rows-per-workgroup
    transforms) = number of polynomials"""
    import struct
    mem = Memory()
    psi, mc = device_tables(64, n, nm, prm)
    c = np.zeros_like(a)
    pa, pb, pc, ppsi, pmc = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(psi), mem.add(mc)
    logn = n.bit_length() - 1
    batch = a.shape[0]
    kernarg = struct.pack("<5Q3i", pc, pa, pb, ppsi, pmc, nm, logn, count if count is not None else 0)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    gx = (batch + 1) // 2 if count is not None else batch << (logn - block_log)
    run_kernel(text, mem, kernarg, (gx, nm), lds, waves_per_wg=(1 << block_log) // 16 // 64)
    out, _ = mem.find(pc, c.nbytes)
    return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()
