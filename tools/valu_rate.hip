// valu_rate.hip -- calibration (dev aid): how many wave64 VALU instructions does an MI355X SIMD issue per
// cycle, per KIND of instruction?  bench.py's `valu_issue` figure prices the scatter's instruction mix against this table.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate [waves per SIMD ...]
// Per instruction kind: every SIMD of the chip runs W waves (default W = 1, 2, 5, 8), each issuing ITER x 32 instructions of
// that kind on eight independent registers (no dependent chain shorter than eight instructions).  Reported per (kind, W):
// wave-instructions per ns chip-wide (HIP events around five launches), the shader clock while the loop ran (s_memtime ticks
// per s_memrealtime tick of 10 ns, one wave's own readings), and from the two the cycles a SIMD spends per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// id, instruction(s), register class (F: one 32-bit register, D: a 64-bit pair), instructions per step
#define OPS(X) \
    X(0, "v_fma_f32 %0, %0, %1, %0", F, 1) \
    X(1, "v_fmac_f32 %0, %1, %1", F, 1) \
    X(2, "v_mul_f32 %0, %0, %1", F, 1) \
    X(3, "v_add_f32 %0, %0, %1", F, 1) \
    X(4, "v_sub_f32 %0, %0, %1", F, 1) \
    X(5, "v_max_f32 %0, %0, %1", F, 1) \
    X(6, "v_add_f32 %0, |%0|, %1", F, 1) \
    X(7, "v_mul_f32 %0, 0x3f800001, %0", F, 1) \
    X(8, "v_pk_fma_f32 %0, %0, %0, %0", D, 1) \
    X(9, "v_pk_mul_f32 %0, %0, %0", D, 1) \
    X(10, "v_pk_add_f32 %0, %0, %0", D, 1) \
    X(11, "v_add_u32 %0, %0, %1", F, 1) \
    X(12, "v_sub_u32 %0, %0, %1", F, 1) \
    X(13, "v_and_b32 %0, %0, %1", F, 1) \
    X(14, "v_or_b32 %0, %0, %1", F, 1) \
    X(15, "v_xor_b32 %0, %0, %1", F, 1) \
    X(16, "v_lshlrev_b32 %0, 1, %0", F, 1) \
    X(17, "v_ashrrev_i32 %0, 1, %0", F, 1) \
    X(18, "v_min_i32 %0, %0, %1", F, 1) \
    X(19, "v_max_u32 %0, %0, %1", F, 1) \
    X(20, "v_mov_b32 %0, %1", F, 1) \
    X(21, "v_cndmask_b32 %0, %0, %1, vcc", F, 1) \
    X(22, "v_cmp_lt_f32 vcc, %0, %1", F, 1) \
    X(23, "v_cmp_le_u32 vcc, %0, %1", F, 1) \
    X(24, "v_cvt_f32_i32 %0, %0", F, 1) \
    X(25, "v_cvt_i32_f32 %0, %0", F, 1) \
    X(26, "v_floor_f32 %0, %0", F, 1) \
    X(27, "v_rndne_f32 %0, %0", F, 1) \
    X(28, "v_med3_f32 %0, %0, %1, %1", F, 1) \
    X(29, "v_mul_u32_u24 %0, %0, %1", F, 1) \
    X(30, "v_mad_u32_u24 %0, %0, %1, %0", F, 1) \
    X(31, "v_mad_i32_i24 %0, %0, %1, %0", F, 1) \
    X(32, "v_mul_lo_u32 %0, %0, %1", F, 1) \
    X(33, "v_mul_hi_u32 %0, %0, %1", F, 1) \
    X(34, "v_mad_u64_u32 %0, vcc, %1, %1, %0", D, 1) \
    X(35, "v_add3_u32 %0, %0, %1, %1", F, 1) \
    X(36, "v_lshl_add_u32 %0, %0, 1, %1", F, 1) \
    X(37, "v_and_or_b32 %0, %0, %1, %1", F, 1) \
    X(38, "v_bfe_u32 %0, %0, 1, 8", F, 1) \
    X(39, "v_perm_b32 %0, %0, %1, %1", F, 1) \
    X(40, "v_min3_i32 %0, %0, %1, %1", F, 1) \
    X(41, "v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", F, 1) \
    X(42, "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf", F, 1) \
    X(43, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", F, 1) \
    X(44, "v_rcp_f32 %0, %0", F, 1) \
    X(45, "v_sqrt_f32 %0, %0", F, 1) \
    X(46, "v_exp_f32 %0, %0", F, 1) \
    X(47, "v_mbcnt_lo_u32_b32 %0, -1, %0", F, 1) \
    X(48, "v_add_co_u32 %0, vcc, %0, %1", F, 1) \
    X(49, "v_div_fixup_f32 %0, %0, %1, %1", F, 1) \
    X(50, "v_div_fmas_f32 %0, %0, %1, %1", F, 1) \
    X(51, "v_div_scale_f32 %0, vcc, %0, %1, %1", F, 1) \
    X(52, "v_fma_f64 %0, %0, %0, %0", D, 1) \
    X(53, "v_add_f64 %0, %0, %0", D, 1) \
    X(54, "v_cvt_f64_f32 %0, %1", D, 1) \
    X(55, "v_cvt_f32_ubyte0 %0, %0", F, 1) \
    X(56, "v_bfi_b32 %0, %0, %1, %1", F, 1) \
    X(57, "v_alignbit_b32 %0, %0, %1, 8", F, 1) \
    X(58, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc", F, 2) \
    X(59, "v_floor_f32 %0, %0\n\tv_cvt_i32_f32 %0, %0", F, 2) \
    X(60, "v_fma_f32 %0, %0, %1, %0\n\tv_lshl_add_u32 %0, %0, 1, %1", F, 2) \
    X(61, "v_readfirstlane_b32 s20, %0", F, 1) \
    X(62, "v_lshlrev_b64 %0, 1, %0", D, 1) \
    X(63, "v_cvt_u32_f32 %0, %0", F, 1) \
    X(64, "v_cndmask_b32 %0, %0, %1, s[22:23]", F, 1) \
    X(65, "v_lshrrev_b32 %0, 1, %0", F, 1) \
    X(66, "v_subrev_u32 %0, %0, %1", F, 1) \
    X(67, "v_not_b32 %0, %0", F, 1) \
    X(68, "v_bcnt_u32_b32 %0, %0, %1", F, 1) \
    X(69, "v_lshlrev_b32 %0, 2, %0", F, 1) \
    X(70, "v_max_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1", F, 2) \
    X(71, "v_cvt_i32_f32 %0, %0\n\tv_add_u32 %0, %0, %1", F, 2) \
    X(72, "v_min_i32 %0, %0, %1\n\tv_and_b32 %0, %0, %1", F, 2) \
    X(73, "v_mad_u32_u24 %0, %0, %1, %0\n\tv_mul_f32 %0, %0, %1", F, 2) \
    X(74, "v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1", F, 2) \
    X(75, "v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1", F, 2) \
    X(76, "v_max_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1", F, 3) \
    X(77, "v_max_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1", F, 3) \
    X(78, "v_cmp_lt_f32 s[22:23], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[22:23]", F, 2) \
    X(79, "v_sub_f32 %0, %1, %0", F, 1) \
    X(80, "v_mul_f32 %0, %0, %1 \n\tv_xor_b32 %0, %0, %1", F, 2) \
    X(81, "v_add_f32 %0, s24, %0", F, 1) \
    X(82, "v_add_u32 %0, 0x12345, %0", F, 1) \
    X(83, "v_fma_f32 %0, %0, %1, %1", F, 1)
constexpr int N_OPS = 84;

constexpr int ITER = 1024;

#define REG_F f[j]
#define REG_D d[j]
template <int OP> __device__ __forceinline__ void step(float (&f)[8], double (&d)[8], float k)
{
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
#define X(id, str, cls, n) if constexpr (OP == id) asm volatile(str : "+v"(REG_##cls) : "v"(k) : "vcc", "s20", "s22", "s23", "s24");
            OPS(X)
#undef X
        }
}

template <int OP> __global__ void __launch_bounds__(256) k_rate(float* sink, unsigned long long* cycles, float k)
{
    float f[8]; double d[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { f[j] = 1.0f + threadIdx.x * 1e-6f + j; d[j] = 1.0 + j + threadIdx.x; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) step<OP>(f, d, k);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) s += f[j] + (float) d[j];
    if (s == 123.456f) sink[0] = s;                 // never true; keeps the registers live
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        cycles[2 * w] = t1 - t0;
        cycles[2 * w + 1] = r1 - r0;
    }
}

typedef void (*Kern)(float*, unsigned long long*, float);
struct OpInfo { const char* name; int per_step; Kern kern; };
static OpInfo kOps[N_OPS] = {
#define X(id, str, cls, n) { str, n, k_rate<id> },
    OPS(X)
#undef X
};

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s: %d CUs, clockRate %.0f MHz (hipDeviceProp); %d x 32 instructions per wave per launch\n", prop.gcnArchName, cus, prop.clockRate / 1e3, ITER);
    int ws[16] = { 1, 2, 5, 8 }, n_ws = 4, op_from = 0;
    if (argc > 1) {
        n_ws = 0;
        for (int i = 1; i < argc && n_ws < 16; i++) {
            if (!strncmp(argv[i], "from=", 5)) op_from = atoi(argv[i] + 5);         // only the kinds from this id on
            else ws[n_ws++] = atoi(argv[i]);
        }
        if (n_ws == 0) { ws[0] = 1; ws[1] = 2; ws[2] = 5; ws[3] = 8; n_ws = 4; }
    }
    float* sink; unsigned long long* cyc;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 32 * 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long* h = (unsigned long long*) malloc(sizeof(unsigned long long) * cus * 32 * 2);
    printf("%-52s %6s %13s %11s %10s %16s\n", "instruction", "w/SIMD", "wave-inst/ns", "us/launch", "clock GHz", "cycles/inst/SIMD");
    for (int op = op_from; op < N_OPS; op++)
        for (int wi = 0; wi < n_ws; wi++) {
            const int w = ws[wi];
            const int blocks = cus * w;             // a workgroup of 256 threads = one wave per SIMD of a CU; w workgroups per CU
            hipLaunchKernelGGL(kOps[op].kern, dim3(blocks), dim3(256), 0, 0, sink, cyc, 1.0000001f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            const int reps = 5;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kOps[op].kern, dim3(blocks), dim3(256), 0, 0, sink, cyc, 1.0000001f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * 4 * 2, hipMemcpyDeviceToHost));
            double ticks = 0.0, real = 0.0;
            for (int i = 0; i < blocks * 4; i++) { ticks += (double) h[2 * i]; real += (double) h[2 * i + 1]; }
            const double ghz = ticks / (real * 10.0);                   // s_memrealtime: 100 MHz
            const double insts_per_wave = (double) ITER * 32 * kOps[op].per_step;
            const double total = insts_per_wave * blocks * 4;
            const double ns = ms * 1e6 / reps;
            // a wave's loop took ticks / waves cycles while its SIMD issued w waves' instructions
            const double cyc_per_inst = ticks / (blocks * 4) / (insts_per_wave * w);
            char name[64];
            strncpy(name, kOps[op].name, 51); name[51] = 0;
            for (char* c = name; *c; c++) if (*c == '\n' || *c == '\t') *c = ' ';
            printf("%-52s %6d %13.1f %11.1f %10.3f %16.3f\n", name, w, total / ns, ns / 1e3, ghz, cyc_per_inst);
        }
    return 0;
}
