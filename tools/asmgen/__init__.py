"""tools/asmgen -- the gfx950 assembly generator behind tools/gen_polymul_asm.py, one module per concern (see each module's
docstring; state.py explains how the register map is shared)."""
