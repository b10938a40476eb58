"""TEST INFRASTRUCTURE: a small interpreter for the subset of gfx950 assembly that the kernel generators
(tools/gen_polymul_asm.py, gen_row1024_u32_asm.py, gen_row128_u16_asm.py, gen_row8_u32_asm.py) emit, so that the
GENERATED kernels can be checked against the oracle on a machine without a GPU (tests/test_asm_emulated.py,
`-m "not gpu"`).

It executes the text the generators produce: one numpy vector of 64 lanes per VGPR, scalar registers, VCC / EXEC / SCC,
a flat "device memory" made of registered buffers (one coherent copy: no caches), one LDS array per workgroup, waves of
a workgroup interleaved at s_barrier, workgroups of a launch either one after the other or -- for the persistent
one-launch plan -- interleaved at their s_sleep polls in an order the test chooses.  Timing, hazards and wait counts are
ignored (every memory operation completes at once); an instruction the interpreter does not know raises, so a generator
change that needs more of the ISA fails loudly here.
"""
import re

import numpy as np

M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1
U = np.uint64
U32 = U(M32)
VCC_LO = 106          # SGPR numbers of VCC (strict-mode bookkeeping only)
STRICT = True
COUNT = False        # tools/asm_cost.py: count executed instructions by class (Wave.counts)


class StrictError(RuntimeError):
    """Strict mode: besides computing results the interpreter keeps, per wave, what real hardware would still have in
    flight, and raises when the executed instruction stream relies on something the ISA does not guarantee:
      * a register read (or overwritten) while a load into it has not been waited for -- counted `s_waitcnt vmcnt(N)` /
        `lgkmcnt(N)` retire the oldest operations first (vector memory and LDS return in order, scalar loads do not);
      * the gfx90a / gfx950 software-visible data hazards the assembler does not pad: VALU writes an SGPR / VCC -> VALU
        reads it (2 wait states), -> a vector-memory instruction reads it (5); VALU writes a VGPR -> v_readfirstlane
        reads it (1); a store of more than 64 bits -> VALU overwrites its data registers (2);
      * LDS words written by one wave and read (or overwritten) by another wave of the workgroup with no s_barrier
        between the two, or an s_barrier reached with LDS writes still outstanding;
      * visibility between workgroups as measured on the MI355X (profiles/r02_l2_flag_probe.txt): plain and sc0 vector
        loads and scalar loads without glc may be served by a cache of the XCD that holds the line from BEFORE another
        workgroup's store; sc1 / nt loads and atomics see memory.  The model is pessimistic: caches never evict (until the
        launch ends) and any two workgroups of an XCD may share a CU."""


class Memory:
    """registered buffers at fake device addresses"""

    def __init__(self):
        self.bufs = []
        self.next = 0x100000000
        self.version = 0          # bumped by every store / atomic (deadlock detection of the concurrent scheduler)
        # strict mode, visibility between workgroups: who may hold a 128-byte line in a cache that is not coherent with
        # other CUs' stores -- line -> {(xcd, workgroup that loaded it): None, or the workgroup whose later store made that
        # copy stale}; one table for the vector L1s, one for the scalar caches; both forgotten at a launch boundary
        self.cur = (0, 0)         # (workgroup, XCD) now running
        self.l1, self.k1 = {}, {}
        self.stale_lines = set()  # lines with at least one stale copy (the only ones a load has to look at closely)

    def new_launch(self):
        self.l1, self.k1 = {}, {}
        self.stale_lines = set()

    def wrote(self, lines):
        me = self.cur[0]
        for line in lines:
            for table, own_ok in ((self.l1, True), (self.k1, False)):
                holders = table.get(line)
                if holders:
                    for key, stale in holders.items():
                        if stale is None and not (own_ok and key[1] == me):   # (a CU's own store updates its own L1)
                            holders[key] = me
                            self.stale_lines.add(line)

    def cached_read(self, table, lines, vector=True):
        """a load that a cache may serve (plain / sc0 vector loads: the CU's L1; scalar loads without glc: the scalar
        cache).  Workgroups of one XCD may or may not share a CU, so a copy that ANY workgroup of this XCD loaded and
        that a store from elsewhere has since made stale is an error -- unless that store was the reader's own (then
        either it updated the copy, same CU, or the reader's CU holds no copy)."""
        me, xcd = self.cur
        for line in lines:
            holders = table.setdefault(line, {})
            for (x, holder), stale in (holders.items() if line in self.stale_lines else ()):
                if x == xcd and stale is not None and not (vector and stale == me):
                    raise StrictError("load of line 0x%x may hit a stale cache line: workgroup %d cached it, workgroup %d stored "
                                      "to it afterwards (needs an sc1 / nt load or an invalidate)" % (line << 7, holder, stale))
            holders[(xcd, me)] = None

    def add(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        base = self.next
        self.bufs.append((base, a))
        self.next += ((a.size + 0xFFFF) & ~0xFFFF) + 0x10000
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
        self.version += 1
        self.wrote(range(addr >> 7, ((addr + len(data) - 1) >> 7) + 1))

    def gather(self, addrs, n, cached=False):
        """addrs: int64 vector -> (len, n) uint8"""
        if cached and STRICT:
            self.cached_read(self.l1, np.unique(np.concatenate((addrs >> 7, (addrs + (n - 1)) >> 7))).tolist())
        a, o = self.find(int(addrs.min()), int(addrs.max() - addrs.min()) + n)
        idx = (addrs - addrs.min() + o)[:, None] + np.arange(n)
        return a[idx]

    def scatter(self, addrs, data):
        n = data.shape[1]
        a, o = self.find(int(addrs.min()), int(addrs.max() - addrs.min()) + n)
        idx = (addrs - addrs.min() + o)[:, None] + np.arange(n)
        a[idx] = data
        self.version += 1
        self.wrote(np.unique(np.concatenate((addrs >> 7, (addrs + (n - 1)) >> 7))).tolist())


# operand kinds
K_V, K_S, K_IMM, K_V2, K_S2, K_VCC, K_EXEC, K_VCCLO, K_OFF, K_VN, K_SN = range(11)
_DEC = {}


def dec(op):
    d = _DEC.get(op)
    if d is None:
        m = re.fullmatch(r"([vs])(\d+)", op)
        m2 = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
        if m:
            d = (K_V if m.group(1) == "v" else K_S, int(m.group(2)))
        elif m2:
            lo, hi = int(m2.group(2)), int(m2.group(3))
            if hi == lo + 1:
                d = (K_V2 if m2.group(1) == "v" else K_S2, lo)
            else:
                d = (K_VN if m2.group(1) == "v" else K_SN, lo)
        elif op == "vcc":
            d = (K_VCC, 0)
        elif op == "exec":
            d = (K_EXEC, 0)
        elif op == "vcc_lo":
            d = (K_VCCLO, 0)
        elif op == "off":
            d = (K_OFF, 0)
        else:
            d = (K_IMM, int(op, 0))
        _DEC[op] = d
    return d


def parse_program(text):
    """-> (instructions, labels): instructions = list of (mnemonic, decoded operands, modifiers, class); reads up to
    .Lfunc_end"""
    ins, labels = [], {}
    pat = r"\b(offset|dst_sel|dst_unused|src0_sel|src1_sel):(\S+)"
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
        for key, val in re.findall(pat, rest):
            mods[key] = int(val, 0) if key == "offset" else val
        rest = re.sub(pat, "", rest)
        for flag in ("sc0", "sc1", "nt", "glc"):
            if re.search(r"\b%s\b" % flag, rest):
                mods[flag] = True
                rest = re.sub(r"\b%s\b" % flag, "", rest)
        if mn == "s_waitcnt":
            vm, lg = re.search(r"vmcnt\((\d+)\)", rest), re.search(r"lgkmcnt\((\d+)\)", rest)
            ops = [int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None]
        elif mn in ("s_nop", "s_sleep", "s_trap"):
            ops = [int(rest.strip() or "0", 0)]
        elif mn == "s_getreg_b32":
            dst, reg = rest.split(",", 1)
            ops = [dec(dst.strip()), reg.strip()]
        elif mn.startswith("s_cbranch") or mn == "s_branch":
            ops = [rest.strip()]
        else:
            ops = [dec(o.strip()) for o in rest.split(",") if o.strip()]
        cls = ("valu" if mn.startswith("v_") else "vmem" if mn.startswith("global_") else "lds" if mn.startswith("ds_") else
               "smem" if mn.startswith("s_load") or mn == "s_memtime" else "salu")
        ins.append((mn, ops, mods, cls))
    return ins, labels


_ALL = np.ones(64, dtype=bool)


def mask_of(bits):
    return np.unpackbits(np.array([bits], dtype="<u8").view(np.uint8), bitorder="little").astype(bool)


def bits_of(flags):
    if np.ndim(flags) == 0:
        return M64 if flags else 0
    return int.from_bytes(np.packbits(flags, bitorder="little").tobytes(), "little")


class Wave:
    ticks = 0      # s_memtime: instructions executed by the whole launch
    counts = {}    # instruction class -> instructions executed by all waves since the table was last cleared (tools/asm_cost.py)

    def __init__(self, prog, labels, mem, lds, kernarg_addr, wg_id, wave_in_wg, wg_y=0, xcc_id=0):
        self.prog, self.labels, self.mem, self.lds = prog, labels, mem, lds
        self.v = np.zeros((256, 64), dtype=np.uint64)     # (kept as uint64, masked to 32 bits)
        self.s = [0] * 108
        self.vcc = 0
        self.scc = 0
        self.pc = 0
        self.set_exec(M64)
        self.s[0], self.s[1] = kernarg_addr & M32, kernarg_addr >> 32
        self.s[2] = wg_id
        self.s[3] = wg_y
        self.v[0] = np.arange(64, dtype=np.uint64) + 64 * wave_in_wg
        self.xcc_id = xcc_id
        # strict mode (see StrictError): what the hardware would still have in flight, and who wrote what when
        self.strict = STRICT
        self.cls = "salu"
        self.slot = 0                 # wait states issued so far
        self.vm_q, self.lgkm_q = [], []           # outstanding operations, oldest first: tuples of destination registers
        self.pend_v, self.pend_s = {}, {}         # register -> number of outstanding operations that will write it
        self.valu_swrite, self.valu_vwrite, self.wide_store = {}, {}, {}   # register -> slot of the producing instruction
        self.wave_in_wg = wave_in_wg
        self.epoch = 0                # barriers passed (LDS race detection)
        self.lds_track = None

    def set_exec(self, bits):
        self.exec = bits & M64
        self.full = self.exec == M64
        self.act = _ALL if self.full else mask_of(self.exec)

    # ---- strict mode
    def fail(self, what):
        mn, ops, mods, cls = self.prog[self.pc - 1]
        raise StrictError("%s (instruction %d: %s, wave %d)" % (what, self.pc - 1, mn, self.wave_in_wg))

    def use_v(self, r, n=1):
        """a VGPR is read by the current instruction"""
        if not self.strict:
            return
        for k in range(r, r + n):
            if self.pend_v and k in self.pend_v:
                self.fail("v%d is read while a load into it is still outstanding (missing s_waitcnt)" % k)
            if self.cls == "valu_lane" and self.slot - self.valu_vwrite.get(k, -99) - 1 < 1:
                self.fail("v%d: VALU write -> v_readfirstlane needs 1 wait state" % k)

    def def_v(self, r, n=1):
        """a VGPR is written by the current instruction"""
        if not self.strict:
            return
        for k in range(r, r + n):
            if self.pend_v and k in self.pend_v:
                self.fail("v%d is written while a load into it is still outstanding" % k)
            if self.cls in ("valu", "valu_lane"):
                if self.slot - self.wide_store.get(k, -99) - 1 < 2:
                    self.fail("v%d: store of more than 64 bits -> VALU write of its data needs 2 wait states" % k)
                self.valu_vwrite[k] = self.slot

    def use_s(self, r, n=1):
        if not self.strict:
            return
        for k in range(r, r + n):
            if self.pend_s and k in self.pend_s:
                self.fail("s%d is read while a scalar load into it is still outstanding" % k)
            gap = self.slot - self.valu_swrite.get(k, -99) - 1
            if self.cls in ("valu", "valu_lane") and gap < 2:
                self.fail("s%d: VALU write -> VALU read needs 2 wait states, has %d" % (k, gap))
            if self.cls == "vmem" and gap < 5:
                self.fail("s%d: VALU write -> VMEM read needs 5 wait states, has %d" % (k, gap))

    def def_s(self, r, n=1):
        if not self.strict:
            return
        for k in range(r, r + n):
            if self.pend_s and k in self.pend_s:
                self.fail("s%d is written while a scalar load into it is still outstanding" % k)
            if self.cls in ("valu", "valu_lane"):
                self.valu_swrite[k] = self.slot
            else:
                self.valu_swrite.pop(k, None)

    def issue(self, queue, dests, kind="v"):
        """a memory operation was issued; dests = registers it will write when it returns"""
        if not self.strict:
            return
        pend = self.pend_s if kind == "s" else self.pend_v
        for k in dests:
            pend[k] = pend.get(k, 0) + 1
        queue.append((kind, tuple(dests)))

    def retire(self, queue, keep):
        while len(queue) > keep:
            kind, dests = queue.pop(0)
            pend = self.pend_s if kind == "s" else self.pend_v
            for k in dests:
                if pend[k] == 1:
                    del pend[k]
                else:
                    pend[k] -= 1

    def waitcnt(self, vm, lgkm):
        if vm is not None:
            self.retire(self.vm_q, vm)            # vector memory operations return in the order issued
        if lgkm is not None:
            # LDS operations return in order; scalar loads may not: with one outstanding only lgkmcnt(0) is a guarantee
            if lgkm == 0 or all(kind != "s" for kind, _ in self.lgkm_q):
                self.retire(self.lgkm_q, lgkm)

    def lds_touch(self, idx, write):
        """LDS race detection: words touched by two waves of the workgroup between the same pair of barriers"""
        if not self.strict:
            return
        t = self.lds_track
        me, ep = self.wave_in_wg, self.epoch
        clash = (t["w_ep"][idx] == ep) & (t["w_wave"][idx] != me)
        if write:
            clash |= (t["r_ep"][idx] == ep) & (t["r_wave"][idx] != me)
        if clash.any():
            self.fail("LDS word %d is %s here and was %s by another wave with no barrier in between"
                      % (int(np.asarray(idx)[clash][0]), "written" if write else "read", "touched" if write else "written"))
        if write:
            t["w_ep"][idx], t["w_wave"][idx] = ep, me
        else:
            same = t["r_ep"][idx] == ep
            t["r_wave"][idx] = np.where(same & (t["r_wave"][idx] != me), -2, me)
            t["r_ep"][idx] = ep

    # ---- operands
    def rd32(self, d):
        k, x = d
        if k == K_V:
            self.use_v(x)
            return self.v[x]
        if k == K_S:
            self.use_s(x)
            return U(self.s[x])
        if k == K_IMM:
            return U(x & M32)
        if k == K_VCCLO:
            self.use_s(VCC_LO)
            return U(self.vcc & M32)
        raise RuntimeError("32-bit operand %r" % (d,))

    def rd64(self, d):
        k, x = d
        if k == K_V2:
            self.use_v(x, 2)
            return self.v[x] | (self.v[x + 1] << U(32))
        if k == K_S2:
            self.use_s(x, 2)
            return U(self.s[x] | (self.s[x + 1] << 32))
        if k == K_IMM:
            return U(x & M64)
        if k == K_VCC:
            self.use_s(VCC_LO, 2)
            return U(self.vcc)
        raise RuntimeError("64-bit operand %r" % (d,))

    def srd(self, d):
        k, x = d
        if k == K_S:
            self.use_s(x)
            return self.s[x]
        if k == K_IMM:
            return x & M32
        if k == K_VCCLO:
            self.use_s(VCC_LO)
            return self.vcc & M32
        raise RuntimeError("scalar operand %r" % (d,))

    def srd64(self, d):
        k, x = d
        if k == K_S2:
            self.use_s(x, 2)
            return self.s[x] | (self.s[x + 1] << 32)
        if k == K_VCC:
            self.use_s(VCC_LO, 2)
            return self.vcc
        if k == K_EXEC:
            return self.exec
        if k == K_IMM:
            return x & M64
        raise RuntimeError("scalar 64-bit operand %r" % (d,))

    def swr(self, d, val):
        assert d[0] == K_S
        self.def_s(d[1])
        self.s[d[1]] = val & M32

    def swr64(self, d, val):
        k, x = d
        if k == K_VCC:
            self.def_s(VCC_LO, 2)
            self.vcc = val & M64
        elif k == K_EXEC:
            self.set_exec(val)
        else:
            assert k == K_S2
            self.def_s(x, 2)
            self.s[x], self.s[x + 1] = val & M32, (val >> 32) & M32

    def wr32(self, d, val):
        assert d[0] == K_V
        self.def_v(d[1])
        if self.full:
            self.v[d[1]] = val & U32
        else:
            np.copyto(self.v[d[1]], val & U32, where=self.act)

    def wr64(self, d, val):
        assert d[0] == K_V2
        self.def_v(d[1], 2)
        if self.full:
            self.v[d[1]] = val & U32
            self.v[d[1] + 1] = val >> U(32)
        else:
            np.copyto(self.v[d[1]], val & U32, where=self.act)
            np.copyto(self.v[d[1] + 1], val >> U(32), where=self.act)

    @staticmethod
    def sel(vec, how):
        if how in (None, "DWORD"):
            return vec
        if how == "WORD_0":
            return vec & U(0xFFFF)
        if how == "WORD_1":
            return (vec >> U(16)) & U(0xFFFF)
        raise RuntimeError("SDWA select %s" % how)

    def gaddr(self, vaddr, saddr, mods):
        """global_* addressing -> int64 byte addresses of the active lanes; `off` = a 64-bit address in a VGPR pair"""
        imm = mods.get("offset", 0)
        if saddr[0] == K_OFF:
            a = self.rd64(vaddr).astype(np.int64) + imm
        else:
            self.use_v(vaddr[1])
            a = self.v[vaddr[1]].astype(np.int64) + (self.srd64(saddr) + imm)
        return a if self.full else a[self.act]

    # ---- execution
    def run(self):
        """generator: yields "barrier" at every s_barrier and "sleep" at every s_sleep, returns at s_endpgm"""
        prog, table = self.prog, HANDLERS
        while True:
            mn, ops, mods, self.cls = prog[self.pc]
            self.pc += 1
            Wave.ticks += 1
            if COUNT:
                Wave.counts[self.cls] = Wave.counts.get(self.cls, 0) + 1
            h = table.get(mn)
            if h is not None:
                h(self, ops, mods)
                self.slot += 1
            elif mn == "s_barrier":
                if self.strict and any(kind == "w" for kind, _ in self.lgkm_q):
                    self.fail("s_barrier with LDS writes of this wave not waited for (the other waves may not see them yet)")
                self.slot += 1
                self.epoch += 1
                yield "barrier"
            elif mn == "s_sleep":
                self.slot += 1
                yield "sleep"
            elif mn == "s_endpgm":
                return
            elif mn == "s_trap":
                raise RuntimeError("s_trap reached (a bounded wait of the kernel ran out)")
            else:
                raise RuntimeError("emulator: unknown instruction %s" % mn)


HANDLERS = {}


def op(*names):
    def reg(fn):
        for n in names:
            HANDLERS[n] = fn
        return fn
    return reg


@op("s_waitcnt")
def _(w, ops, mods):
    if w.strict:
        w.waitcnt(ops[0], ops[1])


@op("s_nop")
def _(w, ops, mods):
    w.slot += ops[0]          # s_nop N = N + 1 wait states (the run loop adds the one)


def _sload(n):
    def h(w, ops, mods):
        addr = w.srd64(ops[1]) + ops[2][1]
        if w.strict and not mods.get("glc"):
            w.mem.cached_read(w.mem.k1, range(addr >> 7, ((addr + 4 * n - 1) >> 7) + 1), vector=False)
        data = np.frombuffer(w.mem.read(addr, 4 * n), dtype=np.uint32)
        lo = ops[0][1]
        if lo % min(n, 4):   # (the assembler rejects it too: destination groups of 2 are even, of 4 and more multiples of 4)
            raise RuntimeError("s_load_dwordx%d into s%d: misaligned destination" % (n, lo))
        w.def_s(lo, n)
        for k in range(n):
            w.s[lo + k] = int(data[k])
        w.issue(w.lgkm_q, range(lo, lo + n), "s")
    return h


for _n, _name in ((1, "s_load_dword"), (2, "s_load_dwordx2"), (4, "s_load_dwordx4"), (8, "s_load_dwordx8"), (16, "s_load_dwordx16")):
    HANDLERS[_name] = _sload(_n)


@op("s_mov_b32")
def _(w, ops, mods):
    w.swr(ops[0], w.srd(ops[1]))


@op("s_mov_b64")
def _(w, ops, mods):
    w.swr64(ops[0], w.srd64(ops[1]))


def _sarith(kind):
    def h(w, ops, mods):
        a, b = w.srd(ops[1]), w.srd(ops[2])
        if kind == "add":
            r = a + b
            w.scc = r >> 32
        elif kind == "addc":
            r = a + b + w.scc
            w.scc = r >> 32
        elif kind == "sub":
            r = a - b
            w.scc = 1 if b > a else 0
        else:
            r = a - b - w.scc
            w.scc = 1 if b + w.scc > a else 0
        w.swr(ops[0], r)
    return h


HANDLERS.update({"s_add_u32": _sarith("add"), "s_addc_u32": _sarith("addc"), "s_sub_u32": _sarith("sub"), "s_subb_u32": _sarith("subb")})


def _s2(fn, sets_scc):
    def h(w, ops, mods):
        r = fn(w.srd(ops[1]), w.srd(ops[2])) & M32
        if sets_scc:
            w.scc = 1 if r else 0
        w.swr(ops[0], r)
    return h


HANDLERS.update({
    "s_lshl_b32": _s2(lambda a, b: a << (b & 31), True), "s_lshr_b32": _s2(lambda a, b: a >> (b & 31), True),
    "s_and_b32": _s2(lambda a, b: a & b, True), "s_or_b32": _s2(lambda a, b: a | b, True),
    "s_xor_b32": _s2(lambda a, b: a ^ b, True), "s_andn2_b32": _s2(lambda a, b: a & ~b, True),
    "s_mul_i32": _s2(lambda a, b: a * b, False), "s_mul_hi_u32": _s2(lambda a, b: (a * b) >> 32, False),
    "s_min_u32": _s2(min, False), "s_max_u32": _s2(max, False),
})


def _signed(x):
    return x - (1 << 32) if x & 0x80000000 else x


def _scmp(fn, signed=False):
    def h(w, ops, mods):
        a, b = w.srd(ops[0]), w.srd(ops[1])
        if signed:
            a, b = _signed(a), _signed(b)
        w.scc = int(fn(a, b))
    return h


HANDLERS.update({
    "s_cmp_lt_u32": _scmp(lambda a, b: a < b), "s_cmp_eq_u32": _scmp(lambda a, b: a == b), "s_cmp_lg_u32": _scmp(lambda a, b: a != b),
    "s_cmp_ge_u32": _scmp(lambda a, b: a >= b), "s_cmp_gt_u32": _scmp(lambda a, b: a > b), "s_cmp_le_u32": _scmp(lambda a, b: a <= b),
    "s_cmp_gt_i32": _scmp(lambda a, b: a > b, True), "s_cmp_lt_i32": _scmp(lambda a, b: a < b, True),
    "s_cmp_ge_i32": _scmp(lambda a, b: a >= b, True), "s_cmp_le_i32": _scmp(lambda a, b: a <= b, True),
    "s_cmp_eq_i32": _scmp(lambda a, b: a == b), "s_cmp_lg_i32": _scmp(lambda a, b: a != b),
})


@op("s_cmp_eq_u64")
def _(w, ops, mods):
    w.scc = int(w.srd64(ops[0]) == w.srd64(ops[1]))


@op("s_cmp_lg_u64")
def _(w, ops, mods):
    w.scc = int(w.srd64(ops[0]) != w.srd64(ops[1]))


def _branch(cond):
    def h(w, ops, mods):
        if cond(w):
            w.pc = w.labels[ops[0]]
    return h


HANDLERS.update({
    "s_branch": _branch(lambda w: True), "s_cbranch_scc1": _branch(lambda w: w.scc == 1), "s_cbranch_scc0": _branch(lambda w: w.scc == 0),
    "s_cbranch_execz": _branch(lambda w: w.exec == 0), "s_cbranch_execnz": _branch(lambda w: w.exec != 0),
    "s_cbranch_vccz": _branch(lambda w: w.vcc == 0), "s_cbranch_vccnz": _branch(lambda w: w.vcc != 0),
})


@op("s_bfe_u32")
def _(w, ops, mods):
    x, c = w.srd(ops[1]), w.srd(ops[2])
    r = (x >> (c & 31)) & ((1 << ((c >> 16) & 0x7F)) - 1)
    w.scc = 1 if r else 0
    w.swr(ops[0], r)


@op("s_lshl_b64")
def _(w, ops, mods):
    r = (w.srd64(ops[1]) << (w.srd(ops[2]) & 63)) & M64
    w.scc = 1 if r else 0
    w.swr64(ops[0], r)


@op("s_cselect_b32")
def _(w, ops, mods):
    w.swr(ops[0], w.srd(ops[1]) if w.scc else w.srd(ops[2]))


@op("s_cselect_b64")
def _(w, ops, mods):
    w.swr64(ops[0], w.srd64(ops[1]) if w.scc else w.srd64(ops[2]))


@op("s_and_saveexec_b64")
def _(w, ops, mods):
    old = w.exec
    w.set_exec(old & w.srd64(ops[1]))
    w.swr64(ops[0], old)
    w.scc = 1 if w.exec else 0


@op("s_ff1_i32_b32")
def _(w, ops, mods):
    x = w.srd(ops[1])
    w.swr(ops[0], (x & -x).bit_length() - 1 if x else M32)


@op("s_not_b32")
def _(w, ops, mods):
    r = ~w.srd(ops[1]) & M32
    w.scc = 1 if r else 0
    w.swr(ops[0], r)


@op("v_bfrev_b32_e32")
def _(w, ops, mods):
    x = w.rd32(ops[1]) & U32
    r = np.zeros(64, dtype=np.uint64)
    for i in range(32):
        r |= ((x >> U(i)) & U(1)) << U(31 - i)
    w.wr32(ops[0], r)


@op("s_getreg_b32")
def _(w, ops, mods):
    if "HW_REG_XCC_ID" not in ops[1]:
        raise RuntimeError("s_getreg_b32 %s" % ops[1])
    w.swr(ops[0], w.xcc_id)


@op("s_memtime")
def _(w, ops, mods):
    w.swr64(ops[0], Wave.ticks)
    w.issue(w.lgkm_q, (ops[0][1], ops[0][1] + 1), "s")


# ---- vector ALU
@op("v_readfirstlane_b32")
def _(w, ops, mods):
    first = (w.exec & -w.exec).bit_length() - 1 if w.exec else 0
    w.cls = "valu_lane"
    w.use_v(ops[1][1])
    w.swr(ops[0], int(w.v[ops[1][1]][first]))


@op("v_mov_b32_e32")
def _(w, ops, mods):
    w.wr32(ops[0], w.rd32(ops[1]))


@op("v_not_b32_e32")
def _(w, ops, mods):
    w.wr32(ops[0], w.rd32(ops[1]) ^ U(0xFFFFFFFF))


@op("v_mad_u64_u32")
def _(w, ops, mods):
    prod = w.rd32(ops[2]) * w.rd32(ops[3])
    r = prod + w.rd64(ops[4])           # (wraps at 64 bits)
    carry = r < prod
    w.swr64(ops[1], bits_of(carry if w.full else (w.act & carry)))
    w.wr64(ops[0], r)


@op("v_mul_u32_u24_sdwa", "v_mul_u32_u24_e32")
def _(w, ops, mods):
    a = w.sel(w.rd32(ops[1]), mods.get("src0_sel")) & U(0xFFFFFF)
    b = w.sel(w.rd32(ops[2]), mods.get("src1_sel")) & U(0xFFFFFF)
    w.wr32(ops[0], a * b)


@op("v_mad_u32_u24")
def _(w, ops, mods):
    w.wr32(ops[0], (w.rd32(ops[1]) & U(0xFFFFFF)) * (w.rd32(ops[2]) & U(0xFFFFFF)) + w.rd32(ops[3]))


@op("v_lshl_add_u32")
def _(w, ops, mods):
    w.wr32(ops[0], (w.rd32(ops[1]) << (w.rd32(ops[2]) & U(31))) + w.rd32(ops[3]))


@op("v_lshl_add_u64")
def _(w, ops, mods):
    w.wr64(ops[0], (w.rd64(ops[1]) << (w.rd32(ops[2]) & U(63))) + w.rd64(ops[3]))


@op("v_bfe_i32")
def _(w, ops, mods):
    x = np.asarray(w.rd32(ops[1]), dtype=np.uint64).astype(np.int64)
    off = np.asarray(w.rd32(ops[2]), dtype=np.uint64).astype(np.int64) & 31
    width = int(np.asarray(w.rd32(ops[3]), dtype=np.uint64).reshape(-1)[0]) & 31
    f = (x >> off) & ((1 << width) - 1)
    f = np.where(f >> (width - 1), f - (1 << width), f) if width else f * 0
    w.wr32(ops[0], (f & M32).astype(np.uint64) + np.zeros(64, dtype=np.uint64))


@op("v_alignbit_b32")
def _(w, ops, mods):
    w.wr32(ops[0], ((w.rd32(ops[1]) << U(32)) | w.rd32(ops[2])) >> (w.rd32(ops[3]) & U(31)))


@op("v_mul_hi_u32")
def _(w, ops, mods):
    w.wr32(ops[0], (w.rd32(ops[1]) * w.rd32(ops[2])) >> U(32))


@op("v_mul_lo_u32")
def _(w, ops, mods):
    w.wr32(ops[0], w.rd32(ops[1]) * w.rd32(ops[2]))


def _vcmp(fn):
    def h(w, ops, mods):
        w.swr64(ops[0], bits_of(w.act & fn(w.rd32(ops[1]), w.rd32(ops[2]))))
    return h


for _c, _f in (("gt", lambda a, b: a > b), ("lt", lambda a, b: a < b), ("ge", lambda a, b: a >= b), ("le", lambda a, b: a <= b),
               ("eq", lambda a, b: a == b), ("ne", lambda a, b: a != b)):
    HANDLERS["v_cmp_%s_u32_e32" % _c] = HANDLERS["v_cmp_%s_u32_e64" % _c] = _vcmp(_f)


@op("v_cndmask_b32_e64", "v_cndmask_b32_e32")
def _(w, ops, mods):
    m = mask_of(w.srd64(ops[3]) if len(ops) > 3 else w.vcc)
    w.wr32(ops[0], np.where(m, w.rd32(ops[2]), w.rd32(ops[1])))


def _v2(fn):
    def h(w, ops, mods):
        w.wr32(ops[0], fn(w.rd32(ops[1]), w.rd32(ops[2])))
    return h


HANDLERS.update({
    "v_add_u32_e32": _v2(lambda a, b: a + b), "v_sub_u32_e32": _v2(lambda a, b: a - b), "v_subrev_u32_e32": _v2(lambda a, b: b - a),
    "v_min_u32_e32": _v2(np.minimum), "v_max_u32_e32": _v2(np.maximum), "v_and_b32_e32": _v2(lambda a, b: a & b),
    "v_or_b32_e32": _v2(lambda a, b: a | b), "v_xor_b32_e32": _v2(lambda a, b: a ^ b),
    "v_lshlrev_b32_e32": _v2(lambda a, b: b << (a & U(31))), "v_lshrrev_b32_e32": _v2(lambda a, b: b >> (a & U(31))),
})


@op("v_ashrrev_i32_e32")
def _(w, ops, mods):
    sh = np.asarray(w.rd32(ops[1]), dtype=np.uint64).astype(np.int64) & 31
    x = np.asarray(w.rd32(ops[2]), dtype=np.uint64).astype(np.int64)
    x = np.where(x & 0x80000000, x - (1 << 32), x)
    w.wr32(ops[0], ((x >> sh) & M32).astype(np.uint64) + np.zeros(64, dtype=np.uint64))


def _vcarry(kind, rev):
    def h(w, ops, mods):
        a, b = w.rd32(ops[2]), w.rd32(ops[3])
        cin = mask_of(w.srd64(ops[4])).astype(np.uint64) if len(ops) > 4 else U(0)
        if rev:
            a, b = b, a
        if kind == "add":
            r = a + b + cin
            cout = r > U32
        else:
            r = a - b - cin            # (wraps in 64 bits; the low word is what counts)
            cout = (b + cin) > a
        w.swr64(ops[1], bits_of(w.act & cout))
        w.wr32(ops[0], r)
    return h


for _sfx in ("e32", "e64"):
    HANDLERS["v_add_co_u32_" + _sfx] = HANDLERS["v_addc_co_u32_" + _sfx] = _vcarry("add", False)
    HANDLERS["v_sub_co_u32_" + _sfx] = HANDLERS["v_subb_co_u32_" + _sfx] = _vcarry("sub", False)
    HANDLERS["v_subrev_co_u32_" + _sfx] = HANDLERS["v_subbrev_co_u32_" + _sfx] = _vcarry("sub", True)


# ---- memory
def _gload(nbytes, signed=False):
    def h(w, ops, mods):
        lo, nreg = ops[0][1], max(1, nbytes // 4)
        w.def_v(lo, nreg)
        if not w.exec:
            w.issue(w.vm_q, range(lo, lo + nreg))
            return
        raw = w.mem.gather(w.gaddr(ops[1], ops[2], mods), nbytes, cached=not (mods.get("sc1") or mods.get("nt")))
        w.issue(w.vm_q, range(lo, lo + nreg))
        if nbytes >= 4:
            words = np.ascontiguousarray(raw).view("<u4")
            for k in range(nbytes // 4):
                _put(w, lo + k, words[:, k].astype(np.uint64))
        elif signed:   # sign-extended to the 32-bit register
            _put(w, lo, np.ascontiguousarray(raw).view("<i2" if nbytes == 2 else np.int8)[:, 0].astype(np.int64).astype(np.uint64) & U32)
        else:
            _put(w, lo, np.ascontiguousarray(raw).view("<u2" if nbytes == 2 else np.uint8)[:, 0].astype(np.uint64))
    return h


def _put(w, r, vals):
    if w.full:
        w.v[r] = vals
    else:
        w.v[r][w.act] = vals


def _get(w, r):
    return w.v[r] if w.full else w.v[r][w.act]


def _gstore(nbytes):
    def h(w, ops, mods):
        lo = ops[1][1]
        w.use_v(lo, max(1, nbytes // 4))
        w.issue(w.vm_q, ())
        if w.strict and nbytes > 8:
            for k in range(lo, lo + nbytes // 4):
                w.wide_store[k] = w.slot
        if not w.exec:
            return
        if nbytes >= 4:
            words = np.stack([_get(w, lo + k) for k in range(nbytes // 4)], axis=1).astype("<u4")
            data = words.view(np.uint8)
        else:
            data = np.ascontiguousarray(_get(w, lo).astype("<u2" if nbytes == 2 else np.uint8)[:, None]).view(np.uint8)
        w.mem.scatter(w.gaddr(ops[0], ops[2], mods), data)
    return h


for _name, _nb in (("dword", 4), ("dwordx2", 8), ("dwordx3", 12), ("dwordx4", 16), ("ushort", 2), ("ubyte", 1)):
    HANDLERS["global_load_" + _name] = _gload(_nb)
HANDLERS["global_load_sbyte"] = _gload(1, True)
HANDLERS["global_load_sshort"] = _gload(2, True)
for _name, _nb in (("dword", 4), ("dwordx2", 8), ("dwordx3", 12), ("dwordx4", 16), ("short", 2), ("byte", 1)):
    HANDLERS["global_store_" + _name] = _gstore(_nb)


def _atomic(fn, width=4):
    """lanes in order (the generated code issues its atomics from lane 0 alone); with a destination (sc0): the old value"""
    def h(w, ops, mods):
        ret = len(ops) == 4
        vaddr, vdata, saddr = (ops[1], ops[2], ops[3]) if ret else (ops[0], ops[1], ops[2])
        w.use_v(vdata[1], width // 4)
        if ret:
            w.def_v(ops[0][1], width // 4)
        w.issue(w.vm_q, range(ops[0][1], ops[0][1] + width // 4) if ret else ())
        lanes = np.nonzero(w.act)[0]
        addrs = w.gaddr(vaddr, saddr, mods)
        mask = (1 << (8 * width)) - 1
        for i, lane in enumerate(lanes):
            d = int(w.v[vdata[1]][lane])
            if width == 8:
                d |= int(w.v[vdata[1] + 1][lane]) << 32
            old = int.from_bytes(w.mem.read(int(addrs[i]), width), "little")
            w.mem.write(int(addrs[i]), (fn(old, d) & mask).to_bytes(width, "little"))
            if ret:
                w.v[ops[0][1]][lane] = old & M32
                if width == 8:
                    w.v[ops[0][1] + 1][lane] = old >> 32
    return h


for _name, _f in (("add", lambda o, d: o + d), ("sub", lambda o, d: o - d), ("or", lambda o, d: o | d), ("and", lambda o, d: o & d),
                  ("xor", lambda o, d: o ^ d), ("swap", lambda o, d: d), ("umax", max), ("umin", min)):
    HANDLERS["global_atomic_" + _name] = _atomic(_f)
    HANDLERS["global_atomic_%s_x2" % _name] = _atomic(_f, 8)


LDS_STATS = {}     # (instruction) -> [wave-instructions, lane-group cycles when conflict-free, extra cycles from bank conflicts]


def lds_bank_cycles(name, byte_addrs, lane_ids):
    """bank model of /opt/skills/guides/MI355X_MICROARCH.md (LDS [CDNA4]): a wave64 access is served in fixed lane groups,
    one LDS cycle each when conflict-free; every extra distinct dword address on a busy bank within a group adds a cycle"""
    kinds = {"ds_read_b32": (32, 32, 1), "ds_write_b32": (32, 32, 1), "ds_read_b64": (32, 64, 2), "ds_write_b64": (16, 32, 2)}
    if name not in kinds:
        return
    group, banks, dwords = kinds[name]
    st = LDS_STATS.setdefault(name, [0, 0, 0])
    st[0] += 1
    for g0 in range(0, 64, group):
        sel = (lane_ids >= g0) & (lane_ids < g0 + group)
        if not sel.any():
            continue
        st[1] += 1
        per_bank = {}
        for a in byte_addrs[sel]:
            for k in range(dwords):
                dw = int(a) // 4 + k
                per_bank.setdefault(dw % banks, set()).add(dw)
        st[2] += max(len(v) for v in per_bank.values()) - 1


def _ds(words, write):
    def h(w, ops, mods):
        imm = mods.get("offset", 0)
        if write:
            w.use_v(ops[0][1])
            w.use_v(ops[1][1], words)
            w.issue(w.lgkm_q, (), "w")
            if not w.exec:
                return
            idx = ((_get(w, ops[0][1]).astype(np.int64) + imm) >> 2)
            if COUNT:
                lds_bank_cycles(w.prog[w.pc - 1][0], idx << 2, np.nonzero(w.act)[0])
            for k in range(words):
                w.lds_touch(idx + k, True)
                w.lds[idx + k] = _get(w, ops[1][1] + k)
        else:
            w.use_v(ops[1][1])
            w.def_v(ops[0][1], words)
            if w.exec:
                idx = ((_get(w, ops[1][1]).astype(np.int64) + imm) >> 2)
                if COUNT:
                    lds_bank_cycles(w.prog[w.pc - 1][0], idx << 2, np.nonzero(w.act)[0])
                vals = [w.lds[idx + k].astype(np.uint64) for k in range(words)]   # (all read before any destination is written)
                for k in range(words):
                    w.lds_touch(idx + k, False)
                    _put(w, ops[0][1] + k, vals[k])
            w.issue(w.lgkm_q, range(ops[0][1], ops[0][1] + words))
    return h


for _name, _nw in (("b32", 1), ("b64", 2), ("b128", 4)):
    HANDLERS["ds_write_" + _name] = _ds(_nw, True)
    HANDLERS["ds_read_" + _name] = _ds(_nw, False)


def _ds_read_i8(w, ops, mods):
    """one byte per lane, sign-extended (the compact rows of tools/gen_row1024_u32_asm.py build_fwd_fma)"""
    imm = mods.get("offset", 0)
    w.use_v(ops[1][1])
    w.def_v(ops[0][1], 1)
    if w.exec:
        addr = _get(w, ops[1][1]).astype(np.int64) + imm
        idx = addr >> 2
        w.lds_touch(idx, False)
        byte = (w.lds[idx].astype(np.uint64) >> ((addr & 3) << 3).astype(np.uint64)) & np.uint64(0xff)
        _put(w, ops[0][1], np.where(byte >= 128, byte | np.uint64(0xffffff00), byte).astype(np.uint64))
    w.issue(w.lgkm_q, range(ops[0][1], ops[0][1] + 1))


HANDLERS["ds_read_i8"] = _ds_read_i8


def _workgroup(waves):
    """generator over one workgroup: its waves advance from barrier to barrier in turn; yields at every s_sleep"""
    nw = len(waves[0].lds)
    track = {"w_ep": np.full(nw, -1, dtype=np.int32), "w_wave": np.full(nw, -1, dtype=np.int16),
             "r_ep": np.full(nw, -1, dtype=np.int32), "r_wave": np.full(nw, -1, dtype=np.int16)}
    for w in waves:
        w.lds_track = track
    gens = [w.run() for w in waves]
    live = list(range(len(waves)))
    while live:
        for i in list(live):
            while True:
                try:
                    ev = next(gens[i])
                except StopIteration:
                    live.remove(i)
                    break
                if ev == "barrier":
                    break
                yield "sleep"      # a poll did not succeed -- let other workgroups run
        yield "barrier"            # every wave is at the barrier (or done): a point where other workgroups may run too


_PARSED = {}


def run_kernel(asm_text, mem, kernarg, grid, lds_bytes, waves_per_wg=4, concurrent=None):
    """execute the workgroups of the kernel in asm_text; grid = gx or (gx, gy); kernarg = bytes.
    concurrent = None: one workgroup after the other (kernels whose workgroups are independent).
    concurrent = (xcc_of, pick): all workgroups resident (persistent kernels that wait for each other): workgroup i runs on
    XCD xcc_of(i); whenever the running workgroup sleeps in a poll or passes a barrier, pick(list of live workgroup numbers)
    names the next."""
    key = hash(asm_text)
    if key not in _PARSED:
        _PARSED[key] = parse_program(asm_text)
    prog, labels = _PARSED[key]
    karg = mem.add(np.frombuffer(kernarg + b"\0" * 64, dtype=np.uint8).copy())
    gx, gy = grid if isinstance(grid, tuple) else (grid, 1)
    mem.new_launch()

    def make(wg):
        lds = np.zeros(lds_bytes // 4 + 16, dtype=np.uint32)
        xcc = concurrent[0](wg) if concurrent else 0
        return _workgroup([Wave(prog, labels, mem, lds, karg, wg % gx, w, wg // gx, xcc) for w in range(waves_per_wg)])

    if concurrent is None:
        for wg in range(gx * gy):
            mem.cur = (wg, 0)
            for _ in make(wg):
                pass               # (a lone workgroup that sleeps just polls again)
        return
    gens = {wg: make(wg) for wg in range(gx * gy)}
    idle, seen = 0, mem.version
    while gens:
        wg = concurrent[1](sorted(gens))
        mem.cur = (wg, concurrent[0](wg))
        try:
            ev = next(gens[wg])
        except StopIteration:
            ev = None
            del gens[wg]
        if mem.version != seen:
            idle, seen = 0, mem.version
        elif ev == "sleep":
            idle += 1
            if idle > 64 * (len(gens) + 1):
                raise RuntimeError("emulated launch is stuck: %d workgroups poll and nothing changes" % len(gens))


def device_tables(limb_bits, n, nm, prm, lane_major=False, incomplete=0):
    """the twiddle table (Tw<T>: {psi^bitrev(k), Shoup companion}) and the ModConst<T> records exactly as
    nfllib_amd/csrc/api.hip build_tables lays them out on the device (negacyclic case), from params<T>.
    lane_major: DevTables::psi_lm, what the ring-mode 64-bit kernels (rows of 8192 words and up) are handed -- the last four stages (indices n/16 .. n-1)
    with stage logn-4+s transposed from [(u << s) + g] to [g * (n/16) + u]"""
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
        if incomplete:
            # DevTables::mc_inc (api.hip build_tables): the records of the incomplete-transform product (tools/asmgen/incomplete.py) --
            # (n / G)^-1 in the n^-1 fields, floor(2^127 / p) - 2^65 in the mu2 field
            g = 1 << incomplete
            ninv_g = ninv * g % p
            w1n_g = w1n * g % p
            rec[3:7] = [ninv_g, (ninv_g << wb) // p, w1n_g, (w1n_g << wb) // p]
            if wb == 64:
                rec[13] = (1 << 127) // p - (1 << 65)
                assert 0 <= rec[13] < (1 << 35)
            elif wb == 32:       # the 32-bit kernels read their Barrett constant from `mu`: floor(2^62 / p) - 2^32 (gen_row1024_u32_asm.py base_mul)
                rec[2] = (1 << 62) // p - (1 << 32)
                assert 0 <= rec[2] < (1 << 32)
        mc[cm] = [r & ((1 << wb) - 1) for r in rec]
    if lane_major:
        assert logn >= 12
        m = n >> 4
        for s_ in range(4):
            lo, cnt = m << s_, m << s_
            blk = psi[:, lo:lo + cnt].reshape(nm, m, 1 << s_, 2)          # [u][g]
            psi[:, lo:lo + cnt] = blk.transpose(0, 2, 1, 3).reshape(nm, cnt, 2)   # [g][u]
    return psi, mc


def run_row_kernel(asm_path, limb_bits, n, nm, prm, a, b, rows_per_wg, with_magic, incomplete=0):
    """a, b: (batch, nm, n) arrays -> the kernel's output array (same shape)"""
    import struct
    mem = Memory()
    psi, mc = device_tables(limb_bits, n, nm, prm, incomplete=incomplete)
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


def run_row_fused(asm_path, n, nm, prm, rows_per_wg, kind, limb_bits=64, incomplete=0, **kw):
    """the transform-fused wave-per-row kernels of tools/asmgen/rows1k.py (64-bit limbs, rows of 1024 / 2048 words) and of
    tools/gen_row1024_u32_asm.py (32-bit limbs, 1024 / 2048 / 4096 words; their forward kinds read the level-2 records: incomplete=2).
    kind "inv": c = INTT(b -+ a k): kw a, b (batch, nm, n) words, key (1 or batch, nm, n) -> c
    kind "fwd": out0 = NTT(x) k0 + NTT(e0) [, out1 = NTT(x) k1 + NTT(e1)]: kw x, e0[, e1]: (B, nm, n) words or (B, n) int8 with B = 1
    (shared by the batch) or batch; k0[, k1] (1 or batch, nm, n) -> out0[, out1]"""
    import struct
    mem = Memory()
    psi, mc = device_tables(limb_bits, n, nm, prm, incomplete=incomplete)
    ppsi, pmc = mem.add(psi), mem.add(mc)
    magic = lambda: 0 if nm == 1 else ((1 << 32) // nm + 1)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    if kind == "inv":
        a, b, key = kw["a"], kw["b"], kw["key"]
        batch = a.shape[0]
        rows = batch * nm
        c = np.zeros_like(a)
        pa, pb, pc, pk = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(key.copy())
        kernarg = struct.pack("<5Q2IQQ2I", pc, pa, pb, ppsi, pmc, nm, magic(), rows, pk, 0 if key.shape[0] == 1 else 1, 0)
        run_kernel(text, mem, kernarg, (rows + rows_per_wg - 1) // rows_per_wg, lds)
        out, _ = mem.find(pc, c.nbytes)
        return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()
    x, e0, k0 = kw["x"], kw["e0"], kw["k0"]
    e1, k1 = kw.get("e1"), kw.get("k1")
    batch = kw["batch"]
    rows = batch * nm
    wdt = np.uint64 if limb_bits == 64 else np.uint32
    o0 = np.zeros((batch, nm, n), dtype=wdt)
    o1 = np.zeros_like(o0)
    stride = lambda arr: 0 if arr.shape[0] == 1 and batch > 1 else 1
    px, pe0, pk0, po0, po1 = mem.add(x.copy()), mem.add(e0.copy()), mem.add(k0.copy()), mem.add(o0), mem.add(o1)
    pe1 = mem.add(e1.copy()) if e1 is not None else pe0
    pk1 = mem.add(k1.copy()) if k1 is not None else pk0
    kernarg = struct.pack("<5Q2I4QQ5I", po0, po1, px, ppsi, pmc, nm, magic(), pk0, pe0, pk1, pe1, rows, stride(x), stride(k0), stride(e0),
                          stride(k1) if k1 is not None else 0, stride(e1) if e1 is not None else 0)
    kernarg += b"\0" * (112 - len(kernarg))
    run_kernel(text, mem, kernarg, (rows + rows_per_wg - 1) // rows_per_wg, lds)
    res = []
    for ptr, arr in ((po0, o0), (po1, o1))[:2 if k1 is not None else 1]:
        buf, _ = mem.find(ptr, arr.nbytes)
        res.append(buf[:arr.nbytes].view(wdt).reshape(arr.shape).copy())
    return res


def run_block_kernel(asm_path, n, nm, prm, a, b, block_log, count=None, words_per_thread=16, grid_x=None, key=None, compact=None,
                     incomplete=0):
    """the 64-bit block kernels of tools/gen_polymul_asm.py (kernarg: dst, a, b, psi, mc, nm, logn[, count]; grid =
    (blocks of the batch, nm); 2^block_log words per workgroup, 16 per thread).  count (the two-rows-per-workgroup
    transforms) = number of polynomials"""
    import struct
    mem = Memory()
    psi, mc = device_tables(64, n, nm, prm, lane_major=block_log >= 13, incomplete=incomplete)   # (the ring-mode kernels: tw_base_lm)
    c = np.zeros_like(a)
    pa, pb, pc, ppsi, pmc = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(psi), mem.add(mc)
    if compact is not None:   # operand a as one signed byte per coefficient, shared by the moduli: (batch, n) int8
        pa = mem.add(compact.copy())
    logn = n.bit_length() - 1
    batch = a.shape[0]
    kernarg = struct.pack("<5Q3i", pc, pa, pb, ppsi, pmc, nm, logn, count if count is not None else 0)
    if grid_x is not None:   # persistent row kernels: workgroup (x, cm) of a (grid_x, nm) grid walks polynomials x, x + grid_x, ...
        kernarg = struct.pack("<5Q4i", pc, pa, pb, ppsi, pmc, nm, logn, batch, grid_x)
    if key is not None:      # the fused inverse kinds of build_row32k: a third input row, one polynomial (stride 0) or one per element
        pk = mem.add(key.copy())
        kernarg = struct.pack("<5Q2iQi", pc, pa, pb, ppsi, pmc, nm, logn, pk, 0 if key.shape[0] == 1 else 1)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    gx = (batch + 1) // 2 if count is not None else batch << (logn - block_log)
    if grid_x is not None:
        gx = grid_x
    run_kernel(text, mem, kernarg, (gx, nm), lds, waves_per_wg=(1 << block_log) // words_per_thread // 64)
    out, _ = mem.find(pc, c.nbytes)
    return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()


def run_row32k_forward_pipeline(asm_path, nm, prm, x8, k0, e0p, k1=None, e1p=None):
    """nflhip_fused_{fma_fwd,enc2_}32768i8_asm (build_row32k): x8 (batch, n) int8; k0 / k1 (1, nm, n) key rows in NTT form; e0p / e1p
    (batch, nm, n) noise rows in NTT form -> out0 [, out1] = NTT(x) k + e'"""
    import struct
    n = 32768
    mem = Memory()
    psi, mc = device_tables(64, n, nm, prm, lane_major=True)
    batch = x8.shape[0]
    o0, o1 = np.zeros_like(e0p), np.zeros_like(e0p)
    two = k1 is not None
    px, pe0, po0, ppsi, pmc = mem.add(x8.copy()), mem.add(e0p.copy()), mem.add(o0), mem.add(psi), mem.add(mc)
    pk0, pk1 = mem.add(k0.copy()), mem.add((k1 if two else k0).copy())
    pe1, po1 = mem.add((e1p if two else e0p).copy()), mem.add(o1)
    kernarg = struct.pack("<5Q2i4Q", po0, px, pe0, ppsi, pmc, nm, 15, pk0, pk1, pe1, po1)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    run_kernel(text, mem, kernarg, (batch, nm), lds, waves_per_wg=16)
    out = []
    for ptr, arr in ((po0, o0), (po1, o1))[:2 if two else 1]:
        buf, _ = mem.find(ptr, arr.nbytes)
        out.append(buf[:arr.nbytes].view(arr.dtype).reshape(arr.shape).copy())
    return out


def run_fused_kernel(asm_path, n, nm, prm, xs, ks, batch, nouts, remap=False, groups=1, lane_major=None):
    """the transform-fused kernels of tools/gen_polymul_asm.py build_fused (kernarg ARGS_FUSED: out0 out1 x0 x1 x2 k0 k1 psi
    mc | nm logn fmt | strides x0 x1 x2 k0 k1 | count magic; grid (batch, nm), or -- remap -- the 1-D grid whose workgroups
    the kernel deals to (batch element, modulus) itself, the nm rows of an element back to back on one XCD).  xs: up to three forward inputs / inverse
    operands -- uint64 arrays (count, nm, n) are word rows (format 0), int8 / int16 / int32 arrays (count, n) the compact
    formats 1 / 2 / 3; ks: key rows (count, nm, n).  An operand with count 1 is shared by the whole batch (stride 0).
    -> list of `nouts` result arrays (batch, nm, n)"""
    import struct
    mem = Memory()
    # (rows of 8192 / 16384 words -- groups = n / 4096 -- and the ring-mode experiment at 4096 read the lane-major twiddle copy)
    psi, mc = device_tables(64, n, nm, prm, lane_major=groups > 1 if lane_major is None else lane_major)
    outs = [np.zeros((batch, nm, n), dtype=np.uint64) for _ in range(nouts)]
    fmt_of = {np.dtype(np.uint64): 0, np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 3}
    px, sx, fmt = [0, 0, 0], [0, 0, 0], 0
    for i, x in enumerate(xs):
        px[i] = mem.add(np.ascontiguousarray(x).copy())
        sx[i] = 0 if x.shape[0] == 1 else 1
        fmt |= fmt_of[x.dtype] << (4 * i)
    pk, sk = [0, 0], [0, 0]
    for i, k in enumerate(ks):
        pk[i] = mem.add(np.ascontiguousarray(k).copy())
        sk[i] = 0 if k.shape[0] == 1 else 1
    po = [mem.add(o) for o in outs] + [0, 0]
    ppsi, pmc = mem.add(psi), mem.add(mc)
    kernarg = struct.pack("<9Q8i2I", po[0], po[1], px[0], px[1], px[2], pk[0], pk[1], ppsi, pmc, nm, n.bit_length() - 1, fmt,
                          sx[0], sx[1], sx[2], sk[0], sk[1], batch, ((1 << 32) // nm + 1) if remap else 0)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    run_kernel(text, mem, kernarg, (nm * 8 * ((batch + 7) // 8), 1) if remap else (batch, nm), lds, waves_per_wg=4 * groups)
    res = []
    for i in range(nouts):
        out, _ = mem.find(po[i], outs[i].nbytes)
        res.append(out[:outs[i].nbytes].view(np.uint64).reshape(outs[i].shape).copy())
    return res


def run_pipe_product(asm_path, n, nm, prm, a, b, one_launch_roles=False, remap=False, b_ntt=False, incomplete=0):
    """n = 32768 / 65536: the three-role kernel of tools/gen_polymul_asm.py build_pipe (kernarg as
    launch_polymul_pipe64k_u64 packs it) driven the way the composed product does: forward streaming pass of both
    operands, fused block products, inverse streaming pass in place -- three launches of the same kernel"""
    import struct
    mem = Memory()
    psi, mc = device_tables(64, n, nm, prm, incomplete=incomplete)
    logn = n.bit_length() - 1
    batch = a.shape[0]
    c, sa, sb = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
    pa, pb, pc, psa, psb = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(sa), mem.add(sb)
    ppsi, pmc = mem.add(psi), mem.add(mc)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    per_row = (28 if logn == 16 else 14) if not b_ntt else 24     # b_ntt (n = 65536 only): b is the transformed operand, read in place

    def launch(cnt_v, cnt_f, cnt_i):
        gx = max(cnt_v, cnt_f, cnt_i) * per_row
        rm = (gx, gx * nm // 8, (1 << 32) // gx + 1) if remap else (0, 0, 0)    # (launch_polymul_pipe64k_u64's XCD remap)
        assert not remap or (gx * nm) % 8 == 0
        kernarg = struct.pack("<5Q5ii5Q2I", pc, psa, pb if b_ntt else psb, ppsi, pmc, nm, logn, cnt_v, cnt_f, cnt_i, rm[0], pa, psa,
                              0 if b_ntt else pb, 0 if b_ntt else psb, pc, rm[1], rm[2])
        run_kernel(text, mem, kernarg, (gx, nm), lds)

    launch(0, batch, 0)
    launch(batch, 0, 0)
    launch(0, 0, batch)
    out, _ = mem.find(pc, c.nbytes)
    return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()


def run_xcd_product(asm_path, n, nm, prm, a, b, dlog, rlog, pooled, wgs, pick, spin=20000, free_mask=0xFF, incomplete=0):
    """n = 32768 / 65536, the whole batch in ONE launch of persistent workgroups (tools/gen_polymul_asm.py fused_header;
    kernarg and work area as launch_polymul_xcd_u64 / k_xcd_reset lay them out).  Workgroup i runs on XCD i mod 8 (the
    hardware's round-robin); `pick` chooses which workgroup continues whenever one sleeps in a poll."""
    import struct
    mem = Memory()
    psi, mc = device_tables(64, n, nm, prm, incomplete=incomplete)
    logn = n.bit_length() - 1
    batch = a.shape[0]
    rows = batch * nm
    assert batch >= 2 and rows >= (8 << dlog)
    assert wgs >= (8 << dlog), "every scheduling domain needs a workgroup (the launcher starts >= 256: 32 per XCD, at most 8 domains)"
    pow2 = (batch & (batch - 1)) == 0
    magic = ((1 << 32) // batch) if pow2 else ((1 << 32) // batch + 1)
    ctl_bytes = 4096 + (8 << dlog) * 0x11000
    slot_bytes = 8 * 32 * n * 8 if pooled else rows * n * 8
    ctl = np.zeros(ctl_bytes, dtype=np.uint8)
    ctl[128:160] = free_mask                              # free masks of the pooled plan (32 slots per XCD)
    scr_a = np.zeros(slot_bytes, dtype=np.uint8)
    scr_b = np.zeros(slot_bytes if not pooled else 16, dtype=np.uint8)
    c = np.zeros_like(a)
    pa, pb, pc, ppsi, pmc = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(psi), mem.add(mc)
    pctl, psa, psb = mem.add(ctl), mem.add(scr_a), mem.add(scr_b)
    kernarg = struct.pack("<5Q4iI5i4Q", pc, pa, pb, ppsi, pmc, nm, logn, rows, batch, magic, dlog, rlog, 0, spin, 0,
                          psa, psb, pctl, 0)
    assert len(kernarg) == 112
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    run_kernel(text, mem, kernarg, wgs, lds, concurrent=(lambda wg: wg % 8, pick))
    out, _ = mem.find(pc, c.nbytes)
    return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()


def run_pipe_product_pipelined(asm_path, n, nm, prm, a, b, incomplete=0):
    """the same three-role kernel driven as the chunked plan drives it: launch t runs the forward pass of polynomial t, the
    block products of polynomial t - 1 and the inverse pass of polynomial t - 2 TOGETHER (all three roles in one launch,
    on different rows), batch + 2 launches in all"""
    import struct
    mem = Memory()
    psi, mc = device_tables(64, n, nm, prm, incomplete=incomplete)
    logn = n.bit_length() - 1
    batch = a.shape[0]
    c, sa, sb = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
    pa, pb, pc, psa, psb = mem.add(a.copy()), mem.add(b.copy()), mem.add(c), mem.add(sa), mem.add(sb)
    ppsi, pmc = mem.add(psi), mem.add(mc)
    with open(asm_path) as f:
        text = f.read()
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    per_row = 28 if logn == 16 else 14
    poly_bytes = nm * n * 8
    for t in range(batch + 2):
        f, v, i = t, t - 1, t - 2
        cnt_f, cnt_v, cnt_i = int(0 <= f < batch), int(0 <= v < batch), int(0 <= i < batch)
        off = lambda base, k: base + max(k, 0) * poly_bytes      # noqa: E731
        kernarg = struct.pack("<5Q6i6Q", off(pc, v), off(psa, v), off(psb, v), ppsi, pmc, nm, logn, cnt_v, cnt_f, cnt_i, 0,
                              off(pa, f), off(psa, f), off(pb, f), off(psb, f), off(pc, i), 0)
        run_kernel(text, mem, kernarg, (max(cnt_v, cnt_f, cnt_i) * per_row, nm), lds)
    out, _ = mem.find(pc, c.nbytes)
    return out[:c.nbytes].view(a.dtype).reshape(a.shape).copy()
