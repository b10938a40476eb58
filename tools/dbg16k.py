#!/usr/bin/env python3
"""Debug driver: one polymul on a 16384-word row, against the 4096-block plan (NFLHIP_ROW16K=0 in a child)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(n, nm, batch):
    import torch
    from nfllib_amd import Engine
    from nfllib_amd.sharding import digest_words
    e = Engine(64, n, nm)
    a = e.fill_uniform(e.empty(batch), 7, 0)
    b = e.fill_uniform(e.empty(batch), 7, 1)
    c = e.polymul(a, b)
    torch.cuda.synchronize()
    h = e.to_host(c)
    print(json.dumps({"digest": digest_words(h), "head": [int(x) for x in h.reshape(-1)[:4]]}))

if sys.argv[1] == "--child":
    child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
else:
    n, nm, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    for v in ("0", "1"):
        env = dict(os.environ, NFLHIP_ROW16K=v)
        r = subprocess.run([sys.executable, __file__, "--child", str(n), str(nm), str(batch)], env=env, capture_output=True, text=True)
        print("ROW16K=%s rc=%d" % (v, r.returncode), r.stdout.strip()[-300:])
        if r.returncode:
            print("\n".join(l for l in r.stderr.splitlines() if "File" not in l)[-1500:])
