/* Test / tooling hooks exported by libpaella_hip.so but NOT part of the public C ABI (include/paella_hip.h).
 * Bound by paella_amd/_lib.py (TEST_HOOKS) for tests/test_gpu_fastmode.py and tools/launch_floor.py only. */
#ifndef PAELLA_TEST_HOOKS_H
#define PAELLA_TEST_HOOKS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* (un)register an arbitrary fp32 [N,K] matrix for the bf16 fast mode so paella_op_gemm(tile_cfg 96..98) can use it */
int paella_test_register_weight(const float* w, size_t numel, int on);
/* launches n_launches dependent, nearly empty kernels (blocks x 256 threads touching n_elems floats): boundary floor */
int paella_test_launch_chain(float* buf, int n_elems, int blocks, int n_launches, void* stream);
/* 1 = run large-query-count attention on the register-fed kernel instead of the LDS-staged one (A/B probe, tools/attn_probe.py) */
int paella_test_attention_variant(int v);
#ifdef __cplusplus
}
#endif
#endif
