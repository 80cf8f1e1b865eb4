// Issue cost of vector instructions on gfx950, measured: cycles one SIMD spends per wave64 instruction when
// its instruction stream is the only thing that limits it (eight waves per SIMD, every wave issuing a long run of
// independent instructions of one kind).  The table prices the instruction mix of the sector-query kernel
// (tools/valu_roofline.py): a VALU-bound kernel cannot finish faster than sum(count x cost) / SIMDs / clock.
//
//   hipcc -O2 --offload-arch=gfx950 tools/valu_issue_bench.hip -o tools/build/valu_issue_bench
//   tools/build/valu_issue_bench > profiles/rNN_valu_issue_cost.json
//
// Method: each wave reads s_memtime (shader clock), runs ITER x 64 instructions, reads s_memtime again.  With W
// waves per SIMD running the same stream, cycles per instruction per SIMD = mean(t1 - t0) / (ITER * 64 * W).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int kIter = 1000;

// 64 independent instructions per loop trip: eight accumulators, eight times each
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

#define KERNEL_F32(name, ASM)                                                                             \
  __global__ __launch_bounds__(256) void name(unsigned long long* out, float seed) {                     \
    float a[8];                                                                                           \
    for (int i = 0; i < 8; ++i) a[i] = seed + (float)(threadIdx.x + i);                                   \
    const float x = seed * 0.999f, y = seed * 1e-3f;                                                      \
    (void)x; (void)y;                                                                                     \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < kIter; ++it) {                                                                  \
      REP64(ASM)                                                                                          \
    }                                                                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
    float s = 0.f;                                                                                        \
    for (int i = 0; i < 8; ++i) s += a[i];                                                                \
    if (s == 12345.678f) out[0] = 1;                                                                      \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                       \
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[1], t1 - t0); atomicAdd(&out[2], r1 - r0); }                  \
  }

#define KERNEL_F64(name, ASM)                                                                             \
  __global__ __launch_bounds__(256) void name(unsigned long long* out, float seed) {                     \
    double a[8];                                                                                          \
    for (int i = 0; i < 8; ++i) a[i] = (double)seed + (double)(threadIdx.x + i);                          \
    const double x = (double)seed * 0.999, y = (double)seed * 1e-3;                                       \
    float b[8];                                                                                           \
    for (int i = 0; i < 8; ++i) b[i] = seed + (float)i;                                                   \
    const float xf = seed * 0.5f;                                                                         \
    (void)x; (void)y; (void)xf;                                                                           \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < kIter; ++it) {                                                                  \
      REP64(ASM)                                                                                          \
    }                                                                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
    double s = 0.;                                                                                        \
    for (int i = 0; i < 8; ++i) s += a[i] + (double)b[i];                                                 \
    if (s == 12345.678) out[0] = 1;                                                                       \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                       \
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[1], t1 - t0); atomicAdd(&out[2], r1 - r0); }                  \
  }

// ---- 32-bit -------------------------------------------------------------------------------------------------
#define A_FMA_F32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define A_ADD_F32(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
#define A_MUL_F32(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_MAX_F32(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_CMP_F32(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
#define A_AND_B32(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_ADD_U32(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_LSHL_B32(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
#define A_MOV_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define A_ADD_DPP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define A_EXP_F32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define A_LOG_F32(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define A_RCP_F32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define A_SQRT_F32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define A_CVT_I32_F32(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define A_CVT_F32_U32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
#define A_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(a[i]));
#define A_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
#define A_READFIRST(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[i]) : "s20");
#define A_CMP_U32(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define A_MOV_B32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(x));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(x) : "s20", "s21");
#define A_LSHL_ADD_U32(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(x));
#define A_MIN_I32(i) asm volatile("v_min_i32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_CMP_LE_F32(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define A_BFE_U32(i) asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(a[i]));
#define A_ADD3_U32(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define A_FMAAK_F32(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7ffff5" : "+v"(a[i]) : "v"(x));
#define A_OR_B32(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_SUB_U32(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_SUB_F32(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_FMAC_F32(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define A_CNDMASK_E32(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
#define A_MUL_LO_U32(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define A_ASHR_I32(i) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
#define A_CVT_SDWA(i) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i]));
#define A_SALU(i) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");

KERNEL_F32(k_fma_f32, A_FMA_F32)
KERNEL_F32(k_add_f32, A_ADD_F32)
KERNEL_F32(k_mul_f32, A_MUL_F32)
KERNEL_F32(k_max_f32, A_MAX_F32)
KERNEL_F32(k_cmp_cndmask_f32, A_CMP_F32)
KERNEL_F32(k_and_b32, A_AND_B32)
KERNEL_F32(k_add_u32, A_ADD_U32)
KERNEL_F32(k_lshlrev_b32, A_LSHL_B32)
KERNEL_F32(k_mov_b32_dpp, A_MOV_DPP)
KERNEL_F32(k_add_u32_dpp, A_ADD_DPP)
KERNEL_F32(k_exp_f32, A_EXP_F32)
KERNEL_F32(k_log_f32, A_LOG_F32)
KERNEL_F32(k_rcp_f32, A_RCP_F32)
KERNEL_F32(k_sqrt_f32, A_SQRT_F32)
KERNEL_F32(k_cvt_i32_f32, A_CVT_I32_F32)
KERNEL_F32(k_cvt_f32_u32, A_CVT_F32_U32)
KERNEL_F32(k_mbcnt, A_MBCNT)
KERNEL_F32(k_readlane, A_READLANE)
KERNEL_F32(k_readfirstlane, A_READFIRST)
KERNEL_F32(k_salu_add, A_SALU)
KERNEL_F32(k_cmp_u32, A_CMP_U32)
KERNEL_F32(k_mov_b32, A_MOV_B32)
KERNEL_F32(k_or_b32, A_OR_B32)
KERNEL_F32(k_sub_u32, A_SUB_U32)
KERNEL_F32(k_sub_f32, A_SUB_F32)
KERNEL_F32(k_fmac_f32, A_FMAC_F32)
KERNEL_F32(k_cndmask_e32, A_CNDMASK_E32)
KERNEL_F32(k_mul_lo_u32, A_MUL_LO_U32)
KERNEL_F32(k_ashr_i32, A_ASHR_I32)
KERNEL_F32(k_cvt_sdwa, A_CVT_SDWA)
KERNEL_F32(k_cndmask_b32, A_CNDMASK)
KERNEL_F32(k_lshl_add_u32, A_LSHL_ADD_U32)
KERNEL_F32(k_min_i32, A_MIN_I32)
KERNEL_F32(k_cmp_le_f32, A_CMP_LE_F32)
KERNEL_F32(k_bfe_u32, A_BFE_U32)
KERNEL_F32(k_add3_u32, A_ADD3_U32)
KERNEL_F32(k_fmaak_f32, A_FMAAK_F32)

// ---- 64-bit -------------------------------------------------------------------------------------------------
#define A_FMA_F64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define A_ADD_F64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[i]) : "v"(y));
#define A_MUL_F64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_MAX_F64(i) asm volatile("v_max_f64 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define A_CMP_F64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define A_CMP_U64(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define A_LSHL_B64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
#define A_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(xf));
#define A_CVT_F32_F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(b[i]) : "v"(x));
#define A_CVT_F64_I32(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(xf));
#define A_RCP_F64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define A_SQRT_F64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define A_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define A_CVT_I32_F64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b[i]) : "v"(x));
#define A_LSHL_ADD_U64(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a[i]) : "v"(x));
#define A_MOV_B64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(x));
#define A_FLOOR_F64(i) asm volatile("v_floor_f64 %0, %0" : "+v"(a[i]));

KERNEL_F64(k_fma_f64, A_FMA_F64)
KERNEL_F64(k_add_f64, A_ADD_F64)
KERNEL_F64(k_mul_f64, A_MUL_F64)
KERNEL_F64(k_max_f64, A_MAX_F64)
KERNEL_F64(k_cmp_f64, A_CMP_F64)
KERNEL_F64(k_cmp_u64, A_CMP_U64)
KERNEL_F64(k_lshlrev_b64, A_LSHL_B64)
KERNEL_F64(k_cvt_f64_f32, A_CVT_F64_F32)
KERNEL_F64(k_cvt_f32_f64, A_CVT_F32_F64)
KERNEL_F64(k_cvt_f64_i32, A_CVT_F64_I32)
KERNEL_F64(k_rcp_f64, A_RCP_F64)
KERNEL_F64(k_sqrt_f64, A_SQRT_F64)
KERNEL_F64(k_floor_f64, A_FLOOR_F64)
KERNEL_F64(k_pk_mul_f32, A_PKMUL)
KERNEL_F64(k_pk_add_f32, A_PKADD)
KERNEL_F64(k_cvt_i32_f64, A_CVT_I32_F64)
KERNEL_F64(k_lshl_add_u64, A_LSHL_ADD_U64)
KERNEL_F64(k_mov_b64, A_MOV_B64)

// ---- LDS (per-wave region, conflict-free) --------------------------------------------------------------------
#define KERNEL_LDS(name, ASM)                                                                             \
  __global__ __launch_bounds__(256) void name(unsigned long long* out, float seed) {                     \
    __shared__ unsigned long long sm[256 * 8];                                                            \
    for (int i = 0; i < 8; ++i) sm[threadIdx.x + 256 * i] = ~0ull;                                        \
    __syncthreads();                                                                                      \
    unsigned long long a[8];                                                                              \
    unsigned int lo[8];                                                                                   \
    for (int i = 0; i < 8; ++i) { a[i] = (unsigned long long)(seed * 1e6f) + threadIdx.x + i; lo[i] = (unsigned int)a[i]; } \
    const unsigned int addr = threadIdx.x * 8u;   /* sm is the only LDS array: offset 0 */                \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < kIter; ++it) {                                                                  \
      REP64(ASM)                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(0)");                                                               \
    }                                                                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
    unsigned long long s = 0;                                                                             \
    for (int i = 0; i < 8; ++i) s += a[i] + lo[i];                                                        \
    if (s == 12345ull) out[0] = 1;                                                                        \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                       \
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[1], t1 - t0); atomicAdd(&out[2], r1 - r0); }                  \
  }
#define A_DS_READ_B32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(lo[i]) : "v"(addr));
#define A_DS_READ_B64(i) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(addr));
#define A_DS_WRITE_B32(i) asm volatile("ds_write_b32 %1, %0" : : "v"(lo[i]), "v"(addr));
#define A_DS_MIN_U64(i) asm volatile("ds_min_u64 %1, %0" : : "v"(a[i]), "v"(addr));
KERNEL_LDS(k_ds_read_b32, A_DS_READ_B32)
KERNEL_LDS(k_ds_read_b64, A_DS_READ_B64)
KERNEL_LDS(k_ds_write_b32, A_DS_WRITE_B32)
KERNEL_LDS(k_ds_min_u64, A_DS_MIN_U64)

typedef void (*kern_t)(unsigned long long*, float);
struct Entry { const char* name; kern_t k; int per_asm; };

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_out = nullptr;
  CK(hipMalloc(&d_out, 8 * sizeof(unsigned long long)));
  unsigned long long h[8];
  const Entry tab[] = {
      {"v_fma_f32", k_fma_f32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_max_f32", k_max_f32, 1},
      {"v_cmp_lt_f32+v_cndmask_b32", k_cmp_cndmask_f32, 2}, {"v_and_b32", k_and_b32, 1},
      {"v_add_u32", k_add_u32, 1}, {"v_lshlrev_b32", k_lshlrev_b32, 1}, {"v_mov_b32_dpp", k_mov_b32_dpp, 1},
      {"v_add_u32_dpp", k_add_u32_dpp, 1}, {"v_exp_f32", k_exp_f32, 1}, {"v_log_f32", k_log_f32, 1}, {"v_rcp_f32", k_rcp_f32, 1},
      {"v_sqrt_f32", k_sqrt_f32, 1}, {"v_cvt_i32_f32", k_cvt_i32_f32, 1}, {"v_cvt_f32_u32", k_cvt_f32_u32, 1},
      {"v_mbcnt_lo_u32_b32", k_mbcnt, 1}, {"v_readlane_b32", k_readlane, 1}, {"v_readfirstlane_b32", k_readfirstlane, 1},
      {"s_add_u32", k_salu_add, 1}, {"v_cmp_lt_u32", k_cmp_u32, 1},
      {"v_or_b32", k_or_b32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_sub_f32", k_sub_f32, 1}, {"v_fmac_f32", k_fmac_f32, 1},
      {"v_cndmask_b32_e32", k_cndmask_e32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_ashrrev_i32", k_ashr_i32, 1},
      {"v_cvt_f32_u32_sdwa", k_cvt_sdwa, 1},
      {"v_mov_b32", k_mov_b32, 1}, {"v_cndmask_b32_e64", k_cndmask_b32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_min_i32", k_min_i32, 1},
      {"v_cmp_le_f32", k_cmp_le_f32, 1}, {"v_bfe_u32", k_bfe_u32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_fmaak_f32", k_fmaak_f32, 1},
      {"v_pk_mul_f32", k_pk_mul_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1}, {"v_cvt_i32_f64", k_cvt_i32_f64, 1},
      {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_mov_b64", k_mov_b64, 1},
      {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_max_f64", k_max_f64, 1},
      {"v_cmp_lt_f64", k_cmp_f64, 1}, {"v_cmp_lt_u64", k_cmp_u64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
      {"v_cvt_f64_f32", k_cvt_f64_f32, 1}, {"v_cvt_f32_f64", k_cvt_f32_f64, 1}, {"v_cvt_f64_i32", k_cvt_f64_i32, 1},
      {"v_rcp_f64", k_rcp_f64, 1}, {"v_sqrt_f64", k_sqrt_f64, 1}, {"v_floor_f64", k_floor_f64, 1},
      {"ds_read_b32", k_ds_read_b32, 1}, {"ds_read_b64", k_ds_read_b64, 1}, {"ds_write_b32", k_ds_write_b32, 1}, {"ds_min_u64", k_ds_min_u64, 1},
  };
  // Eight waves per SIMD resident (256-thread blocks, eight per CU), four rounds of them so that the uneven start
  // and end of a grid weigh little.  The SIMDs of the chip issue total_instructions in wall time; the shader clock
  // during the run is the ratio of the two in-kernel counters (s_memtime: shader cycles; s_memrealtime: 100 MHz).
  const int rounds = 4;
  printf("{\n \"device\": \"%s\", \"compute_units\": %d, \"simds\": %d,\n \"method\": \"%d blocks of 256 threads (8 resident per CU, %d rounds), "
         "%d x 64 independent instructions per wave; cycles per wave64 instruction per SIMD = wall time x shader clock x SIMDs / instructions issued; "
         "shader clock = sum(delta s_memtime) / sum(delta s_memrealtime) x 100 MHz over all waves\",\n \"cycles_per_wave64_instruction_per_simd\": {\n",
         prop.gcnArchName, cus, cus * 4, cus * 8 * rounds, rounds, kIter);
  bool first = true;
  for (const Entry& e : tab) {
    const int blocks = cus * 8 * rounds;
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_out, 0, 8 * sizeof(unsigned long long)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f);
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipMemcpy(h, d_out, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double ghz = (double)h[1] / (double)h[2] * 0.1;
    const double n_inst = (double)blocks * 4 * kIter * 64 * e.per_asm;
    const double cyc = (double)ms * 1e-3 * ghz * 1e9 * (cus * 4) / n_inst;
    // the same from the waves' own clocks: a wave's delta covers its instructions and those of the waves it shares the SIMD with
    const double cyc_wave = (double)h[1] / ((double)blocks * 4) / ((double)kIter * 64 * e.per_asm) / 8.0;
    printf("%s  \"%s\": {\"cycles\": %.3f, \"from_wave_clocks\": %.3f, \"kernel_ms\": %.3f, \"shader_clock_GHz\": %.3f}", first ? "" : ",\n", e.name, cyc,
           cyc_wave, ms, ghz);
    first = false;
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  }
  printf("\n }\n}\n");
  CK(hipFree(d_out));
  return 0;
}
