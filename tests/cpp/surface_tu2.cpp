// Second translation unit including the whole header: the link must succeed with no
// duplicate symbols (the reference's tests/multi0.cpp + multi1.cpp hygiene test).
#include <nfl_hip/nfl.hpp>

int other_tu_selftest() {
  using poly_t = nfl::poly<uint64_t, 64, 3>;
  void *mem = nullptr;
  if (posix_memalign(&mem, 32, sizeof(poly_t) * 2) != 0) return 3;
  poly_t *p = new (mem) poly_t(nfl::uniform(9));
  poly_t *q = new (p + 1) poly_t(*p);
  q->ntt_pow_phi();
  q->invntt_pow_invphi();
  const bool ok = !bool(*p != *q);
  free(mem);
  return ok ? 0 : 1;
}
