// Microbenchmark: issue cost of wave64 vector instructions on MI355X (gfx950) as a function of the waves resident on a
// SIMD.  Settles the question behind DESIGN.md section 4: does a SIMD retire one wave64 VALU instruction every 4 cycles
// (one wave alone) or every 2 (MI355X_MICROARCH.md, constants table) for the instruction mix of k_select?
//
// Every kernel runs ITERS x 64 instructions of one kind per wave (8 independent register chains x 8 per asm block,
// so neither dependency latency nor the loop overhead matters) and stamps s_memtime around the loop.  The grid is
// 256 CUs x w workgroups of 256 threads: one wave per SIMD and workgroup, hence w waves per SIMD (w = 1, 2, 3, 4, 6, 8).
// Every wave records its s_memtime start/end and its (XCC, SE, SH, CU, SIMD) from HW_REG_HW_ID / HW_REG_XCC_ID, so the
// waves that REALLY shared a SIMD are known (the dispatcher does not place exactly w per SIMD).
// Reported: cycles per instruction PER SIMD = median over SIMDs of (last end - first start of its waves) / (instructions
// those waves issued), and the SIMDs seen : min-max waves per SIMD observed.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip        Run: ./valu_issue > out.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define ITERS 8192

// (xcc, se, sh, cu, simd) of the running wave: HW_REG_HW_ID (id 4) bits simd[5:4] cu[11:8] sh[12] se[15:13], HW_REG_XCC_ID (id 20)
__device__ __forceinline__ unsigned long long hw_id() {
    unsigned a, b;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(a), "=s"(b));
    return ((unsigned long long)(b & 0xf) << 16) | (a & 0xfff0u);
}

// one asm statement = 8 instructions on 8 independent registers; the C loop repeats it 8 x ITERS times
#define KERNEL_F32(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float x, float y) {               \
        float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;  \
        __syncthreads();                                                                                   \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < ITERS; ++it) {                                                               \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                  \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(x), "v"(y) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                                            \
        }                                                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                       \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            const size_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                \
            out[3 * wi] = t1 - t0; out[3 * wi + 1] = t0; out[3 * wi + 2] = hw_id();                        \
        }                                                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;                              \
    }

#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define OPS2(op, s2) op " %0, " s2 ", %0\n" op " %1, " s2 ", %1\n" op " %2, " s2 ", %2\n" op " %3, " s2 ", %3\n" \
                     op " %4, " s2 ", %4\n" op " %5, " s2 ", %5\n" op " %6, " s2 ", %6\n" op " %7, " s2 ", %7\n"
#define OPS(op, sfx) op " %0, %0 " sfx "\n" op " %1, %1 " sfx "\n" op " %2, %2 " sfx "\n" op " %3, %3 " sfx "\n" \
                     op " %4, %4 " sfx "\n" op " %5, %5 " sfx "\n" op " %6, %6 " sfx "\n" op " %7, %7 " sfx "\n"

KERNEL_F32(k_fma, OP3("v_fma_f32"))
KERNEL_F32(k_mul, OP2("v_mul_f32"))
KERNEL_F32(k_add, OP2("v_add_f32"))
KERNEL_F32(k_muladd, "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                     "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n")
KERNEL_F32(k_and, OP2("v_and_b32"))
KERNEL_F32(k_bcnt, OP2("v_bcnt_u32_b32"))
KERNEL_F32(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                      "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL_F32(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                  "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
KERNEL_F32(k_cndmask_s, "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                        "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n")
KERNEL_F32(k_cndmask_x, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                        "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
KERNEL_F32(k_cmp_s, "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                    "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n")
KERNEL_F32(k_mbcnt, OPS2("v_mbcnt_lo_u32_b32", "s20"))
KERNEL_F32(k_sub_s, OPS2("v_sub_f32", "s20"))
KERNEL_F32(k_sub, OP2("v_sub_f32"))
KERNEL_F32(k_mul_s, OPS2("v_mul_f32", "s20"))
KERNEL_F32(k_fma_s, "v_fma_f32 %0, s20, %0, %8\n v_fma_f32 %1, s20, %1, %8\n v_fma_f32 %2, s20, %2, %8\n v_fma_f32 %3, s20, %3, %8\n"
                    "v_fma_f32 %4, s20, %4, %8\n v_fma_f32 %5, s20, %5, %8\n v_fma_f32 %6, s20, %6, %8\n v_fma_f32 %7, s20, %7, %8\n")
KERNEL_F32(k_fmac, OP2("v_fmac_f32"))
KERNEL_F32(k_fmac_s, OPS2("v_fmac_f32", "s20"))
KERNEL_F32(k_mul_c, OPS2("v_mul_f32", "0x40490fdb"))
KERNEL_F32(k_mul_i, OPS2("v_mul_f32", "2.0"))
KERNEL_F32(k_and_s, OPS2("v_and_b32", "s20"))
KERNEL_F32(k_or, OP2("v_or_b32"))
KERNEL_F32(k_addu, OP2("v_add_u32"))
KERNEL_F32(k_lshl, OPS2("v_lshlrev_b32", "2"))
KERNEL_F32(k_cndmask_w, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL_F32(k_cndmask_p, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
KERNEL_F32(k_lshl_add, OP3("v_lshl_add_u32"))
KERNEL_F32(k_add3, OP3("v_add3_u32"))
KERNEL_F32(k_max, OP2("v_max_f32"))
KERNEL_F32(k_cvt, OP1("v_cvt_i32_f32"))
KERNEL_F32(k_mul_lo, OP2("v_mul_lo_u32"))
KERNEL_F32(k_sqrt, OP1("v_sqrt_f32"))
KERNEL_F32(k_rcp, OP1("v_rcp_f32"))
KERNEL_F32(k_exp, OP1("v_exp_f32"))
KERNEL_F32(k_log, OP1("v_log_f32"))
KERNEL_F32(k_dpp, OPS("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL_F32(k_dpp_add, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                      "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                      "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
// LDS crossbar shuffles: 8 in flight, then the wait (issue rate of the LDS pipe, not the latency)
KERNEL_F32(k_bpermute, "ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                       "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n")
// a DEPENDENT chain of shuffles (what a 6-step binary search over lane values costs): 8 steps, each waited for
KERNEL_F32(k_bpermute_dep, "ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n"
                           "ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n"
                           "ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n"
                           "ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)\n")
KERNEL_F32(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                       "v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3\n")

// 64-bit register operands
#define KERNEL_F64(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, float xf, float yf) {             \
        double x = xf, y = yf;                                                                             \
        double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7; \
        __syncthreads();                                                                                   \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < ITERS; ++it) {                                                               \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                  \
                asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(x), "v"(y) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                                            \
        }                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            const size_t wi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                \
            out[3 * wi] = t1 - t0; out[3 * wi + 1] = t0; out[3 * wi + 2] = hw_id();                        \
        }                                                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = 0;                               \
    }
KERNEL_F64(k_fma64, OP3("v_fma_f64"))
KERNEL_F64(k_mul64, OP2("v_mul_f64"))
KERNEL_F64(k_add64, OP2("v_add_f64"))
KERNEL_F64(k_pk_mul, OP2("v_pk_mul_f32"))
KERNEL_F64(k_pk_add, OP2("v_pk_add_f32"))
KERNEL_F64(k_pk_fma, OP3("v_pk_fma_f32"))

typedef void (*kern_t)(unsigned long long*, float, float);
struct Entry { const char* name; kern_t k; };

int main() {
    const Entry tab[] = {
        {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_mul+v_add mix", k_muladd}, {"v_and_b32", k_and},
        {"v_bcnt_u32_b32", k_bcnt}, {"v_mbcnt_lo_u32_b32 (sgpr mask)", k_mbcnt}, {"v_sub_f32 (sgpr operand)", k_sub_s}, {"v_sub_f32", k_sub}, {"v_mul_f32 (sgpr operand)", k_mul_s}, {"v_fma_f32 (sgpr operand)", k_fma_s}, {"v_fmac_f32", k_fmac},
        {"v_fmac_f32 (sgpr operand)", k_fmac_s}, {"v_mul_f32 (32-bit literal)", k_mul_c}, {"v_mul_f32 (inline constant)", k_mul_i}, {"v_and_b32 (sgpr operand)", k_and_s},
        {"v_or_b32", k_or}, {"v_add_u32", k_addu}, {"v_lshlrev_b32 (inline const)", k_lshl},
        {"v_cmp->vcc then 7 v_cndmask vcc", k_cndmask_w}, {"v_cmp->vcc, v_cndmask vcc pairs", k_cndmask_p}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3},
        {"v_max_f32", k_max}, {"v_cvt_i32_f32", k_cvt}, {"v_mul_lo_u32", k_mul_lo},
        {"v_cndmask_b32 (vcc, dst=src0)", k_cndmask}, {"v_cndmask_b32 (sgpr pair mask)", k_cndmask_s}, {"v_cndmask_b32 (vcc, dst!=src)", k_cndmask_x}, {"v_cmp_lt_f32 -> vcc", k_cmp}, {"v_cmp_lt_f32 -> sgpr pairs", k_cmp_s}, {"v_pk_mul_f32", k_pk_mul},
        {"v_pk_add_f32", k_pk_add}, {"v_pk_fma_f32", k_pk_fma}, {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64},
        {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log}, {"v_mov_b32_dpp", k_dpp},
        {"v_add_f32_dpp", k_dpp_add}, {"v_readlane_b32", k_readlane}, {"ds_bpermute_b32 (8 in flight)", k_bpermute},
        {"ds_bpermute_b32 (dependent)", k_bpermute_dep},
    };
    const int ws[] = {1, 2, 3, 4, 6, 8};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs; %d x 64 instructions per wave; cycles per instruction per SIMD (median SIMD: span of its waves / their instructions) (SIMDs seen : min-max waves per SIMD)\n",
           prop.name, cus, ITERS);
    printf("%-32s", "instruction");
    for (int w : ws) printf("   w=%d            ", w);
    printf("\n");
    unsigned long long* d;
    hipMalloc(&d, (size_t)cus * 8 * 4 * 8 * 3);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Entry& e : tab) {
        printf("%-32s", e.name);
        for (int w : ws) {
            const int grid = cus * w, nw = grid * 4;
            double best_cpi = 1e30, clk = 0; size_t nsimd = 0, wlo = 0, whi = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> h3((size_t)nw * 3);
                hipMemcpy(h3.data(), d, (size_t)nw * 24, hipMemcpyDeviceToHost);
                // waves per SIMD as placed by the dispatcher; cycles per instruction of a SIMD = the span of its waves
                // (first start .. last end) / instructions they issued together
                std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> simd;
                unsigned long long tmin = ~0ull, tmax = 0;
                for (int i = 0; i < nw; ++i) {
                    const unsigned long long dt = h3[3 * i], t0 = h3[3 * i + 1], id = h3[3 * i + 2];
                    simd[id].push_back({t0, t0 + dt});
                    tmin = std::min(tmin, t0); tmax = std::max(tmax, t0 + dt);
                }
                std::vector<double> cp;
                size_t wmin = 1000, wmax = 0;
                for (auto& kv : simd) {
                    unsigned long long a = ~0ull, b = 0;
                    for (auto& p : kv.second) { a = std::min(a, p.first); b = std::max(b, p.second); }
                    cp.push_back((double)(b - a) / ((double)ITERS * 64.0 * kv.second.size()));
                    wmin = std::min(wmin, kv.second.size()); wmax = std::max(wmax, kv.second.size());
                }
                std::sort(cp.begin(), cp.end());
                const double cpi = cp[cp.size() / 2];
                if (cpi < best_cpi) { best_cpi = cpi; clk = (double)(tmax - tmin) / (ms * 1e-3) / 1e9; nsimd = simd.size(); wlo = wmin; whi = wmax; }
            }
            printf("  %5.2f (%zu:%zu-%zu)", best_cpi, nsimd, wlo, whi); (void)clk;
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
