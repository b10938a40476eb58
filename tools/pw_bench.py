import sys, os
sys.path.insert(0, "/root/repo")
import torch
from nfllib_amd import Engine, OP_ADD, OP_MUL
e = Engine(64, 4096, 4)
batch = 16384
a = e.fill_uniform(e.empty(batch), 1, 0); b = e.fill_uniform(e.empty(batch), 1, 1); c = e.empty(batch)
def rate(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
by = 3 * batch * 4 * 4096 * 8
print("add %.2f TB/s  mul %.2f TB/s" % (by / rate(lambda: e.pointwise(OP_ADD, a, b, out=c)) / 1e12, by / rate(lambda: e.pointwise(OP_MUL, a, b, out=c)) / 1e12))
d = e.fill_uniform(e.empty(batch), 2, 0)
print("eval a*b+d %.2f TB/s" % (4 * batch * 4 * 4096 * 8 / rate(lambda: e.eval([0, 1, 0x12, 2, 0x10], [a, b, d], out=c)) / 1e12),
      "fill %.2f TB/s" % (batch * 4 * 4096 * 8 / rate(lambda: e.fill_uniform(c, 3, 0)) / 1e12),
      "any_neq %.2f TB/s" % (2 * batch * 4 * 4096 * 8 / rate(lambda: e.any_neq(a, b)) / 1e12))
key = e.fill_uniform(e.empty(1), 5, 0)
out = e.empty(batch)
print("strided eval u*key+e %.2f TB/s" % (3 * batch * 4 * 4096 * 8 / rate(lambda: e.eval_strided([0, 1, 0x12, 2, 0x10], [a, key, d], [1, 0, 1], out, batch=batch)) / 1e12))
