// pt_kernels_fma.hip — the OPT-IN contracted instantiation of the restart kernel (PTAMD_KERNEL_BVH_RESTART_FMA).
//
// The default kernels execute the reference's IEEE operation sequence with nothing fused (-ffp-contract=off) and equal the oracle
// bit for bit.  The reference's own binary does not: it is built with nvcc's default --fmad=true (cuda_opengl/CMakeLists.txt:20-22).
// This translation unit compiles the SAME source with contraction allowed — the compiler fuses a * b + c wherever it likes, as
// nvcc may — into a second set of kernels in namespace ptamd_fma.  Its images are NOT bit-identical to the oracle's; they are
// held to the measured tolerance between faithful builds (tests/test_gpu_parity.py:
// test_contracted_kernel_stays_inside_the_stated_tolerance) and reported as `value_fma` beside `value`, never as the headline.
// Only the path-tracing kernel is contracted: the resolve pass (accumulate, tonemap, store) is the exact one.
#pragma clang fp contract(fast)
#define PT_FMA_BUILD 1
#ifndef PT_ASM_LEAF
#define PT_ASM_LEAF 0   /* the hand-scheduled triangle test is the exact operation sequence: here the compiled, contracted one runs */
#endif
#define ptamd ptamd_fma
#include "pt_kernels.hip"
