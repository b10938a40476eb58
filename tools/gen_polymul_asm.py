#!/usr/bin/env python3
"""Generate nfllib_amd/csrc/polymul4096_gfx950.s -- the hand-scheduled gfx950 assembly
version of the metric kernel (fused c = INTT(NTT(a) (.) NTT(b)), u64, n = 4096).

Same algorithm, thread mapping, LDS layout and device tables as k_polymul4096 in
kernels_fast.hip (read that file's header first); what the generator adds over hipcc:

  * every v_mad_u64_u32 addend pair is placed by construction (even-aligned VGPR pairs,
    a persistent zero register behind the one zero-extended operand), so the ~1000
    v_mov copies per wave that hipcc needs to build those pairs disappear;
  * the 64-bit accumulate chains write straight into the coefficient registers
    (no result moves), the Shoup low word and the "- (q << 62)" term are one
    four-instruction v_mad_u64_u32 chain + one v_add_u32;
  * two independent butterflies are interleaved instruction by instruction, which
    covers the 2 wait states gfx950 needs between a VALU SGPR write (carry / borrow)
    and the VALU that consumes it; a hazard tracker inserts s_nop where it does not;
  * a's and b's forward passes share each pass's twiddle registers.

The generator is the package tools/asmgen (state.py: register map and configuration; emitter.py; arith.py: the 62-bit
butterflies; twiddles.py; block4096.py: the metric kernel and its siblings; rows.py / rows32k.py: 8192 ... 32768-word rows; pipe.py / xcd.py: n = 65536;
fused.py: the transform-fused pipelines; objfile.py; main.py: which kernels, under which register map).  This file is its command
line and what the three small-row generators import (Emitter, interleave, HEADER, FOOTER, args_yaml, ROOT).

Run:  python tools/gen_polymul_asm.py   (writes the .s; nfllib_amd/csrc/Makefile assembles it
with clang -x assembler -mcpu=gfx950, links it with ld.lld and embeds the code object).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmgen.emitter import Emitter, interleave          # noqa: E402,F401
from asmgen.main import main                            # noqa: E402
from asmgen.objfile import FOOTER, HEADER, args_yaml    # noqa: E402,F401
from asmgen.state import ROOT                           # noqa: E402,F401

if __name__ == "__main__":
    main()
