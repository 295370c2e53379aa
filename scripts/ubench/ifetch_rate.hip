// Micro-benchmark: is the instruction supply (one instruction cache per two CUs) a limit for VALU-dense code on gfx950?
// The same dependent-free stream of wave64 VALU instructions in encodings of different size, with and without scalar
// instructions between them, 8 waves per SIMD, loop bodies of 1 KB - 16 KB (well inside the cache):
//   e32   v_add_f32_e32          4 bytes per VALU instruction
//   e64   v_add_f32_e64          the same operation in the 8-byte VOP3 encoding
//   fma   v_fma_f32              8 bytes
//   lit   v_add_f32_e32 + literal 8 bytes (VOP2 with a 32-bit literal)
//   e32+s / e64+s                one s_mov_b32 (4 bytes) after every VALU instruction: twice the instructions, 8 / 12 bytes per pair
// If the wave-instruction rate follows bytes per instruction rather than instruction count, fetch bandwidth is the roof.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/ifetch_rate.hip -o build/ifetch_rate && build/ifetch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REGS "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "s40", "s41", "s42", "s43", "s44", "s45", "vcc", "scc"
// eight independent accumulators v16..v23, operands v24 / v25 (set by the kernel), 128 VALU instructions per block of text
#define BODY8(OP)  OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23)
#define BODY128(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) \
                    BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP)
#define OP_E32(r)   "v_add_f32_e32 v" #r ", v24, v" #r "\n\t"
#define OP_E64(r)   "v_add_f32_e64 v" #r ", v24, v" #r "\n\t"
#define OP_FMA(r)   "v_fma_f32 v" #r ", v" #r ", v24, v25\n\t"
#define OP_MADMK(r) "v_add_f32_e32 v" #r ", 0x3f800347, v" #r "\n\t"
#define OP_E32S(r)  "v_add_f32_e32 v" #r ", v24, v" #r "\n\ts_mov_b32 s40, s41\n\t"
#define OP_E64S(r)  "v_add_f32_e64 v" #r ", v24, v" #r "\n\ts_mov_b32 s40, s41\n\t"
#define OP_FMAS(r)  "v_fma_f32 v" #r ", v" #r ", v24, v25\n\ts_mov_b32 s40, s41\n\t"
#define OP_E32NOP(r)  "v_add_f32_e32 v" #r ", v24, v" #r "\n\ts_nop 0\n\t"
#define OP_E32WAIT(r) "v_add_f32_e32 v" #r ", v24, v" #r "\n\ts_waitcnt vmcnt(0)\n\t"
#define OP_E32AND(r)  "v_add_f32_e32 v" #r ", v24, v" #r "\n\ts_and_b64 s[42:43], s[44:45], s[44:45]\n\t"
#define OP_E32S2(r)   "v_add_f32_e32 v" #r ", v24, v" #r "\n\ts_mov_b32 s40, s41\n\ts_mov_b32 s42, s41\n\t"
#define OP_E32BR(r)   "v_add_f32_e32 v" #r ", v24, v" #r "\n\ts_cbranch_scc1 1f\n\t1:\n\t"   /* scc = 0: never taken */
#define OP_2E32S(r)   "v_add_f32_e32 v" #r ", v24, v" #r "\n\tv_add_f32_e32 v" #r ", v25, v" #r "\n\ts_mov_b32 s40, s41\n\t"
#define OP_4E32S(r)   "v_add_f32_e32 v" #r ", v24, v" #r "\n\tv_add_f32_e32 v" #r ", v25, v" #r "\n\tv_add_f32_e32 v" #r ", v24, v" #r "\n\tv_add_f32_e32 v" #r ", v25, v" #r "\n\ts_mov_b32 s40, s41\n\t"
#define OP_E32DS(r)   "v_add_f32_e32 v" #r ", v24, v" #r "\n\tv_add_f32_e32 v" #r ", v25, v" #r "\n\tv_add_f32_e32 v" #r ", v24, v" #r "\n\tv_add_f32_e32 v" #r ", v25, v" #r "\n\tds_read_b32 v26, v27\n\t"
#define OP_PKFMA(r)   "v_pk_fma_f32 v[" #r ":" #r "+1], v[" #r ":" #r "+1], v[24:25], v[26:27]\n\t"
#define OP_PKFMAB(r)  "v_pk_fma_f32 v[" #r ":" #r "+1], v[" #r ":" #r "+1], v[24:25], v[26:27] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define OP_SONLY(r)   "s_mov_b32 s40, s41\n\t"
#define OP_CMP(r)     "v_add_f32_e32 v" #r ", v24, v" #r "\n\tv_cmp_lt_f32_e32 vcc, v24, v" #r "\n\t"

template <int MODE, int BLOCKS128>
__global__ void __launch_bounds__(256) stream(float* out, int iters)
{
  float acc = 0.f;
  asm volatile("v_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 0.5\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\t"
               "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\ts_mov_b32 s41, 0" ::: REGS, "v24", "v25");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int b = 0; b < BLOCKS128; ++b) {
      if (MODE == 0) asm volatile(BODY128(OP_E32) ::: REGS);
      if (MODE == 1) asm volatile(BODY128(OP_E64) ::: REGS);
      if (MODE == 2) asm volatile(BODY128(OP_FMA) ::: REGS);
      if (MODE == 3) asm volatile(BODY128(OP_MADMK) ::: REGS);
      if (MODE == 4) asm volatile(BODY128(OP_E32S) ::: REGS);
      if (MODE == 5) asm volatile(BODY128(OP_E64S) ::: REGS);
      if (MODE == 6) asm volatile(BODY128(OP_FMAS) ::: REGS);
      if (MODE == 7) asm volatile(BODY128(OP_E32NOP) ::: REGS);
      if (MODE == 8) asm volatile(BODY128(OP_E32WAIT) ::: REGS);
      if (MODE == 9) asm volatile(BODY128(OP_E32AND) ::: REGS);
      if (MODE == 10) asm volatile(BODY128(OP_E32S2) ::: REGS);
      if (MODE == 11) asm volatile("s_cmp_eq_u32 s41, 1\n\t" BODY128(OP_E32BR) ::: REGS);
      if (MODE == 12) {   // even waves: VALU only; odd waves: SALU only (do different waves' VALU and SALU instructions issue together?)
        if ((threadIdx.x >> 6) & 1) asm volatile(BODY128(OP_SONLY) ::: REGS); else asm volatile(BODY128(OP_E32) ::: REGS);
      }
      if (MODE == 13) asm volatile(BODY128(OP_CMP) ::: REGS);
      if (MODE == 17) asm volatile("v_mov_b32 v26, 0.5\n\tv_mov_b32 v27, 0.5\n\t" OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22) OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22)
                                   OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22) OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22)
                                   OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22) OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22)
                                   OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22) OP_PKFMA(16) OP_PKFMA(18) OP_PKFMA(20) OP_PKFMA(22) ::: REGS, "v26", "v27");
      if (MODE == 18) asm volatile("v_mov_b32 v26, 0.5\n\tv_mov_b32 v27, 0.5\n\t" OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22) OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22)
                                   OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22) OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22)
                                   OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22) OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22)
                                   OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22) OP_PKFMAB(16) OP_PKFMAB(18) OP_PKFMAB(20) OP_PKFMAB(22) ::: REGS, "v26", "v27");
      if (MODE == 14) asm volatile(BODY128(OP_2E32S) ::: REGS);
      if (MODE == 15) asm volatile(BODY128(OP_4E32S) ::: REGS);
      if (MODE == 16) asm volatile("v_mov_b32 v27, 0\n\t" BODY128(OP_E32DS) "s_waitcnt lgkmcnt(0)\n\t" ::: REGS, "v26", "v27");
    }
  }
  asm volatile("v_add_f32 %0, v16, v17\n\tv_add_f32 %0, %0, v18\n\tv_add_f32 %0, %0, v19\n\tv_add_f32 %0, %0, v20" : "=v"(acc) :: REGS);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int B>
static void run(const char* name, int bytes_valu, int bytes_pair, float* out, int waves_per_simd)
{
  const int blocks = 256 * waves_per_simd, iters = 40000 / B;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((stream<MODE, B>), dim3(blocks), dim3(256), 0, 0, out, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, B>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double valu = (double)blocks * 4 * iters * B * 128;
  const double bytes = valu * (bytes_pair ? bytes_pair : bytes_valu);
  // 128 instruction caches (one per two CUs)
  printf("%-8s body %5d B  %d waves/SIMD  %8.3f ms  %7.1f G VALU wave-inst/s  %7.1f G inst/s  %6.1f B/ns per instruction cache\n", name,
         B * 128 * (bytes_pair ? bytes_pair : bytes_valu), waves_per_simd, best, valu / best / 1e6, valu * (bytes_pair ? 2 : 1) / best / 1e6, bytes / best / 1e6 / 128.0);
}

int main()
{
  float* out;
  hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float));
  for (int w : {8}) {
    run<0, 2>("e32", 4, 0, out, w);
    run<1, 2>("e64", 8, 0, out, w);
    run<2, 2>("fma", 8, 0, out, w);
    run<3, 2>("lit", 8, 0, out, w);
    run<4, 2>("e32+s", 4, 8, out, w);
    run<5, 2>("e64+s", 8, 12, out, w);
    run<6, 2>("fma+s", 8, 12, out, w);
  }
  for (int w : {8, 2}) {
    run<7, 2>("e32+nop", 4, 8, out, w);
    run<8, 2>("e32+wait", 4, 8, out, w);
    run<9, 2>("e32+and", 4, 8, out, w);
    run<10, 2>("e32+2s", 4, 12, out, w);   // (the "G inst/s" column counts two per VALU: three here, x1.5)
    run<11, 2>("e32+br", 4, 8, out, w);
    run<12, 2>("split", 4, 0, out, w);     // (per wave: 128 VALU OR 128 SALU per block; the VALU column counts both kinds)
    run<13, 2>("e32+cmp", 4, 8, out, w);
    run<14, 2>("2e32+s", 12, 0, out, w);    // (counted per group: 2 VALU + 1 SALU = one "instruction" of 12 bytes in the columns)
    run<15, 2>("4e32+s", 20, 0, out, w);    // (4 VALU + 1 SALU per group)
    run<16, 2>("4e32+ds", 24, 0, out, w);   // (4 VALU + 1 ds_read_b32 per group)
  }
  // packed fp32: 32 v_pk_fma_f32 per block (the columns count a block as 128 instructions: multiply the rates by 0.25)
  for (int w : {8, 4, 2, 1}) { run<2, 2>("fma", 8, 0, out, w); run<17, 8>("pkfma/4", 8, 0, out, w); run<18, 8>("pkfmaB/4", 8, 0, out, w); }
  // body size: 1 KB ... 16 KB of fma
  run<2, 1>("fma", 8, 0, out, 8);
  run<2, 8>("fma", 8, 0, out, 8);
  run<2, 16>("fma", 8, 0, out, 8);
  run<0, 16>("e32", 4, 0, out, 8);
  return 0;
}
