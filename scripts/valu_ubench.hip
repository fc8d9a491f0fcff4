// valu_ubench.hip -- issue cost of the VALU instruction kinds K1 (sparse_align.hip) is made of, on gfx950.
//
// For every kind: W waves per SIMD (1, 2, 4) each run a loop of 8 independent instructions x 256 rounds;
// the shader clock (s_memtime) around the loop gives cycles per wave-instruction seen by one wave, and
// x / W the SIMD's issue cost per instruction once enough waves hide the dependent latency.  Used to
// check what SQ_INSTS_VALU and SQ_ACTIVE_INST_VALU mean for bench.py's roofline_valu.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/valu_ubench.hip -o build/valu_ubench && build/valu_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define ROUNDS 256

// 8 independent destinations per round; operands chosen so values stay finite
#define OP8_F32(INS)                                                                                      \
  asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
               INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
               : "v"(a), "v"(b))
#define OP8_2(INS)                                                                                        \
  asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"              \
               INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"              \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])  \
               : "v"(a))
#define OP8_1(INS)                                                                                        \
  asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" \
               INS " %6, %6\n" INS " %7, %7\n"                                                            \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]))
#define OP8_F64(INS)                                                                                      \
  asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
               INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])  \
               : "v"(da), "v"(db))
#define OP8_F64_2(INS)                                                                                    \
  asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"              \
               INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"              \
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])  \
               : "v"(da))
#define OP8_F64_1(INS)                                                                                    \
  asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" \
               INS " %6, %6\n" INS " %7, %7\n"                                                            \
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]))

enum Kind {
  FMA_F32, MUL_F32, ADD_F32, FMA_F64, MUL_F64, ADD_F64, CNDMASK, CVT_UBYTE, ALIGNBYTE, CVT_F32_F64, CVT_F64_F32,
  RCP_F32, RSQ_F32, RCP_F64, FLOOR_F32, CVT_I32_F32, AND_B32, LSHR_B32, MUL_LO_U32, MOV_DPP, CMP_F32, READLANE,
  DEP_FMA_F32, DEP_FMA_F64, CNDMASK_E64, CNDMASK_VCC_ONES, BFI_B32, PERM_B32, FMAC_F32, PK_FMA_F32, PK_MUL_F32,
  PK_ADD_F32, N_KINDS
};
static const char* kind_name[N_KINDS] = {
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cndmask_b32", "v_cvt_f32_ubyte0",
    "v_alignbyte_b32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_rcp_f32", "v_rsq_f32", "v_rcp_f64", "v_floor_f32",
    "v_cvt_i32_f32", "v_and_b32", "v_lshrrev_b32", "v_mul_lo_u32", "v_mov_b32 dpp row_shr:1", "v_cmp_lt_f32 (vcc)",
    "v_readlane_b32", "v_fma_f32 dependent chain", "v_fma_f64 dependent chain", "v_cndmask_b32_e64 (sgpr pair)",
    "v_cndmask_b32 vcc=-1", "v_bfi_b32", "v_perm_b32", "v_fmac_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};

template <int KIND>
__global__ void __launch_bounds__(256) bench(long long* cycles, float* sink) {
  float r[8];
  double d[8];
  for (int k = 0; k < 8; ++k) {
    r[k] = 1.0f + 0.001f * (float)(threadIdx.x + k);
    d[k] = 1.0 + 0.001 * (double)(threadIdx.x + k);
  }
  float a = 0.999f, b = 0.001f;
  double da = 0.999, db = 0.001;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(da), "+v"(db));
  __syncthreads();
  const long long t0 = (long long)__builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ROUNDS; ++it) {
    if (KIND == FMA_F32) OP8_F32("v_fma_f32");
    if (KIND == MUL_F32) OP8_2("v_mul_f32");
    if (KIND == ADD_F32) OP8_2("v_add_f32");
    if (KIND == FMA_F64) OP8_F64("v_fma_f64");
    if (KIND == MUL_F64) OP8_F64_2("v_mul_f64");
    if (KIND == ADD_F64) OP8_F64_2("v_add_f64");
    if (KIND == CNDMASK) asm volatile(
        "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
        "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a) : "vcc");
    if (KIND == CVT_UBYTE) OP8_1("v_cvt_f32_ubyte0");
    if (KIND == ALIGNBYTE) OP8_F32("v_alignbyte_b32");
    if (KIND == CVT_F32_F64) asm volatile(
        "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
        "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
        : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
    if (KIND == CVT_F64_F32) asm volatile(
        "v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
        "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
    if (KIND == RCP_F32) OP8_1("v_rcp_f32");
    if (KIND == RSQ_F32) OP8_1("v_rsq_f32");
    if (KIND == RCP_F64) OP8_F64_1("v_rcp_f64");
    if (KIND == FLOOR_F32) OP8_1("v_floor_f32");
    if (KIND == CVT_I32_F32) OP8_1("v_cvt_i32_f32");
    if (KIND == AND_B32) OP8_2("v_and_b32");
    if (KIND == LSHR_B32) asm volatile(
        "v_lshrrev_b32 %0, 1, %0\nv_lshrrev_b32 %1, 1, %1\nv_lshrrev_b32 %2, 1, %2\nv_lshrrev_b32 %3, 1, %3\n"
        "v_lshrrev_b32 %4, 1, %4\nv_lshrrev_b32 %5, 1, %5\nv_lshrrev_b32 %6, 1, %6\nv_lshrrev_b32 %7, 1, %7\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
    if (KIND == MUL_LO_U32) OP8_2("v_mul_lo_u32");
    if (KIND == MOV_DPP) asm volatile(
        "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
    if (KIND == CMP_F32) asm volatile(
        "v_cmp_lt_f32 vcc, %0, %1\nv_cmp_lt_f32 vcc, %1, %2\nv_cmp_lt_f32 vcc, %2, %3\nv_cmp_lt_f32 vcc, %3, %4\n"
        "v_cmp_lt_f32 vcc, %4, %5\nv_cmp_lt_f32 vcc, %5, %6\nv_cmp_lt_f32 vcc, %6, %7\nv_cmp_lt_f32 vcc, %7, %0\n"
        : : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]) : "vcc");
    if (KIND == READLANE) asm volatile(
        "v_readlane_b32 s20, %0, 1\nv_readlane_b32 s21, %1, 2\nv_readlane_b32 s22, %2, 3\nv_readlane_b32 s23, %3, 4\n"
        "v_readlane_b32 s24, %4, 5\nv_readlane_b32 s25, %5, 6\nv_readlane_b32 s26, %6, 7\nv_readlane_b32 s27, %7, 8\n"
        : : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (KIND == CNDMASK_E64) asm volatile(
        "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\n"
        "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\nv_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\n"
        "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a) : "s20", "s21");
    if (KIND == CNDMASK_VCC_ONES) asm volatile(
        "s_mov_b64 vcc, -1\n"
        "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
        "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a) : "vcc");
    if (KIND == BFI_B32) OP8_F32("v_bfi_b32");
    if (KIND == PERM_B32) OP8_F32("v_perm_b32");
    if (KIND == FMAC_F32) OP8_2("v_fmac_f32");
    if (KIND == PK_FMA_F32) OP8_F64("v_pk_fma_f32");  // (two f32 per register pair)
    if (KIND == PK_MUL_F32) OP8_F64_2("v_pk_mul_f32");
    if (KIND == PK_ADD_F32) OP8_F64_2("v_pk_add_f32");
    if (KIND == DEP_FMA_F32) asm volatile(
        "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
        "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
        : "+v"(r[0]) : "v"(a), "v"(b));
    if (KIND == DEP_FMA_F64) asm volatile(
        "v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
        "v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\n"
        : "+v"(d[0]) : "v"(da), "v"(db));
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += r[k] + (float)d[k];
  if (s == 123.456f) sink[0] = s;  // keeps the registers alive
  if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run_kind(long long* d_cycles, float* d_sink, int n_cu) {
  printf("  \"%s\": {", kind_name[KIND]);
  for (int wi = 0; wi < 3; ++wi) {
    const int W = 1 << wi;  // waves per SIMD: W workgroups of 4 waves per CU
    const int blocks = n_cu * W;
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), d_cycles, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (size_t i = 0; i < h.size(); ++i) sum += (double)h[i];
    const double per_wave_instr = sum / (double)h.size() / (8.0 * ROUNDS);
    printf("%s\"waves_per_simd_%d\": {\"cycles_per_instr_seen_by_a_wave\": %.2f, \"simd_cycles_per_instr\": %.2f}", wi ? ", " : "",
           W, per_wave_instr, per_wave_instr / W);
  }
  printf("}%s\n", KIND + 1 < N_KINDS ? "," : "");
}

template <int K>
struct Runner {
  static void go(long long* c, float* s, int n) {
    run_kind<K>(c, s, n);
    Runner<K + 1>::go(c, s, n);
  }
};
template <>
struct Runner<N_KINDS> {
  static void go(long long*, float*, int) {}
};

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
  const int n_cu = p.multiProcessorCount;
  long long* d_cycles;
  float* d_sink;
  hipMalloc(&d_cycles, sizeof(long long) * (size_t)n_cu * 4 * 4);
  hipMalloc(&d_sink, 4);
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"note\": \"one workgroup of 4 waves per CU and per wave-per-SIMD step; "
         "the hardware scheduler is assumed to spread them evenly\",\n", p.gcnArchName, n_cu);
  Runner<0>::go(d_cycles, d_sink, n_cu);
  printf("}\n");
  return 0;
}
