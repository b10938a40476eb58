/* nflhip_debug.h -- test hooks of libnflhip.so.  NOT part of the product boundary (include/nflhip.h): nothing in
 * include/nfl_hip/nfl.hpp, nfllib_amd/ or bench.py calls these; tests do (tests/test_gpu_xcd.py, tests/test_gpu_samplers.py).
 * They exist so that the library reads no test knobs from the environment. */
#ifndef NFLHIP_DEBUG_H
#define NFLHIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* number of launches of the one-launch plan (persistent workgroups, rows of 32768 / 65536 words) issued by this process:
 * lets a test assert WHICH plan served a call */
unsigned long long nflhip_debug_xcd_launches(void);

/* the Gaussian samplers compare the top 64 bits of a uniform number with the top word of a table entry and fetch the
 * number's lower words only on a tie (2^-64 per entry).  shift > 0 makes the comparison treat words that agree in their
 * top (64 - shift) bits as ties, so that the tie path runs on nearly every sample -- the VALUE drawn does not change.
 * 0 restores production behaviour. */
void nflhip_debug_gauss_tie_shift(int shift);

/* where the pipelined host-pointer path of this context spent its time so far (seconds): out[0] host threads copying
 * caller memory into the pinned slots, out[1] copying results out, out[2] waiting for the device, out[3] inside the calls */
struct nflhip_ctx;
void nflhip_debug_host_pipe_seconds(const struct nflhip_ctx *ctx, double out[4]);

/* variants of the transform-fused kernels: 0 = the library's policy (forward entries with compact inputs and more than one
 * modulus deal the nm rows of a batch element to one XCD, everything else runs one workgroup per (element, modulus) in a
 * 2-D grid; at degree 4096 the inverse entries run on the ring-mode register map, the forward ones on the pair-mode map),
 * 1 = always the 2-D grid, 2 = always the XCD-dealt 1-D grid, 3 = the other register map at degree 4096, 4 = rows of 1024 / 2048 (/ 4096
 * at 32-bit limbs) words run the compiled one-pass template (kernels_wave.hip k_row_fwd_fma / k_row_fma_inv) instead of the generated
 * wave-per-row kernels: their same-box A/B partner.  Same results. */
void nflhip_debug_fused_grid(int mode);

/* which kernel serves nflhip_polymul[_dev] at 64-bit limbs, degree 4096, coefficient-form operands: 0 = complete transforms,
 * 1 / 2 = the forward transforms stop that many stages early, the products are taken modulo X^2 / X^4 -+ zeta and the inverse
 * starts as many stages late (tools/asmgen/incomplete.py).  Same words out.  Returns the previous setting; a negative level
 * only reads it.  Process-wide. */
int nflhip_debug_polymul_level(int level);

/* role trace of the one-launch plan (rows of 32768 / 65536 words, NFLHIP_XCD): device memory of 32 x 65536 x 16 bytes, zeroed by the
 * caller; every role of the following products stores {ticket | kind << 28, t0, t1, t2} = when its workgroup became free, when the
 * role's inputs were ready, when it was done (low words of s_memtime) at record (domain << 16 | sequence number in the domain).
 * NULL switches the trace off.  tools/xcd_trace.py turns a launch into the table of profiles/r06_E_role_stamps.txt. */
void nflhip_debug_xcd_trace(void *device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* NFLHIP_DEBUG_H */
