"""CPU: the generated gfx950 listings are pinned.  tools/asmgen/listing_sha256.json holds the SHA-256 of every .s file the four
generators write (the 64 kernels of tools/gen_polymul_asm.py = the package tools/asmgen, and the small-row generators'); a
refactoring of the generator must leave every digest unchanged (how the round-5 split of the 4 300-line single file into the
package was accepted), a deliberate kernel change updates the manifest in the same commit:

    python tools/gen_polymul_asm.py && python tools/gen_row1024_u32_asm.py && python tools/gen_row128_u16_asm.py && \
        python tools/gen_row8_u32_asm.py && python tests/test_asm_listings.py --update
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nfllib_amd", "csrc")
MANIFEST = os.path.join(ROOT, "tools", "asmgen", "listing_sha256.json")
GENERATORS = ("gen_polymul_asm.py", "gen_row1024_u32_asm.py", "gen_row128_u16_asm.py", "gen_row8_u32_asm.py")


def _digests():
    return {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in sorted(glob.glob(os.path.join(CSRC, "*_gfx950.s")))}


def test_generators_reproduce_the_pinned_listings():
    env = {k: v for k, v in os.environ.items() if not k.startswith("NFL_GEN_") and k != "NFL_DEBUG16K"}   # (no ablation / experiment switches)
    for g in GENERATORS:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", g)], stdout=subprocess.DEVNULL, env=env)
    want = json.load(open(MANIFEST))
    got = _digests()
    assert set(want) <= set(got), sorted(set(want) - set(got))
    changed = sorted(k for k in want if got[k] != want[k])
    assert not changed, "generated listings differ from tools/asmgen/listing_sha256.json: %s" % changed
    # the one listing kept in history for review is the generator's output too
    assert hashlib.sha256(open(os.path.join(CSRC, "polymul4096nt_gfx950.s"), "rb").read()).hexdigest() == want["polymul4096nt_gfx950.s"]


def test_the_package_has_no_module_level_state_outside_state_py():
    """every value configure() / main() rebinds lives in asmgen/state.py and is read as cfg.NAME: a `from .state import X` elsewhere
    would freeze X at import time"""
    import re
    pkg = os.path.join(ROOT, "tools", "asmgen")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f not in ("state.py", "__init__.py"):
            text = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^from \.state import", text, flags=re.M), f
            assert "globals()" not in text, f


if __name__ == "__main__":
    if "--update" in sys.argv:
        json.dump(_digests(), open(MANIFEST, "w"), indent=0, sort_keys=True)
        print("wrote", MANIFEST)
