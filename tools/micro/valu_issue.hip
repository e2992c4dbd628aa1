// Diagnostic (not product): issue cost of vector instruction classes on gfx950 with the SIMDs fully occupied.
// Each wave runs ITER x 16 independent instructions of one class; the report is SIMD cycles per wave-instruction at
// the clock the chip held (s_memtime deltas of one wave per workgroup are averaged as a cross-check of wall time).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/micro/valu_issue.hip ; run: ./valu_issue [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP16(X) X X X X X X X X X X X X X X X X
#define ITER 4096

template <int K>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, ib = 3;
    double d0 = threadIdx.x, d1 = d0 + 1, db = 1.0000001;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (K == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(b));) }
        if (K == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));) }
        if (K == 2) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(ib)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(ib)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 3) { REP16(asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b) : "vcc");) }
        if (K == 4) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i0) : "v"(ib)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i2) : "v"(ib)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 5) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(db)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d1) : "v"(db)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(db)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d1) : "v"(db));) }
        if (K == 6) { REP16(asm volatile("v_max_f32 %0, |%0|, %1" : "+v"(a0) : "v"(b)); asm volatile("v_max_f32 %0, |%0|, %1" : "+v"(a1) : "v"(b)); asm volatile("v_max_f32 %0, |%0|, %1" : "+v"(a2) : "v"(b)); asm volatile("v_max_f32 %0, |%0|, %1" : "+v"(a3) : "v"(b));) }
        if (K == 7) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(i0) : "v"(ib)); asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(i1)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i2) : "v"(ib)); asm volatile("v_mov_b32 %0, %1" : "+v"(i3) : "v"(i0));) }
        if (K == 8) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d0) : "v"(db)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d1) : "v"(db)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d0) : "v"(db)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d1) : "v"(db));) }
        if (K == 9) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 10) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a0)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a1)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a2)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a3));) }
        if (K == 11) { REP16(asm volatile("v_cmp_lt_f32 vcc, %1, %0" :: "v"(a0), "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %1, %0" :: "v"(i1), "v"(ib) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %0" :: "v"(a2), "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %1, %0" :: "v"(i3), "v"(ib) : "vcc");) }
        if (K == 12) { REP16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a3) : "v"(b));) }
        if (K == 13) { REP16(asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a1) : "v"(b)); asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a3) : "v"(b));) }
        if (K == 14) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b) : ); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b));) }
        if (K == 15) { REP16(asm volatile("v_cmp_lt_f32 s[20:21], %1, %0" :: "v"(a0), "v"(b) : "s20", "s21"); asm volatile("v_cmp_lt_f32 s[22:23], %1, %0" :: "v"(a1), "v"(b) : "s22", "s23"); asm volatile("v_cmp_lt_f32 s[20:21], %1, %0" :: "v"(a2), "v"(b) : "s20", "s21"); asm volatile("v_cmp_lt_f32 s[22:23], %1, %0" :: "v"(a3), "v"(b) : "s22", "s23");) }
        if (K == 16) { REP16(asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(i0) : "v"(ib) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(i1) :: "vcc"); asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(i2) : "v"(ib) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(i3) :: "vcc");) }
        if (K == 17) { REP16(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i0)); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i1)); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i2)); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i3));) }
        if (K == 18) { REP16(asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(i0) : "v"(ib)); asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(i2) : "v"(ib)); asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 19) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(i0) : "v"(ib)); asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(i2) : "v"(ib)); asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 20) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(b));) }
        if (K == 21) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i0) : "v"(ib)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i2) : "v"(ib)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 22) { REP16(asm volatile("v_min_u32 %0, %0, %1" : "+v"(i0) : "v"(ib)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(i1) : "v"(ib)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(i2) : "v"(ib)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(i3) : "v"(ib));) }
        if (K == 23) { REP16(asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(i0), "v"(ib) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(i1) :: "vcc"); asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(i2), "v"(ib) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(i3) :: "vcc");) }
        if (K == 24) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a1) : "v"(b)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a3) : "v"(b));) }
        if (K == 25) { REP16(asm volatile("v_cvt_f32_i32 %0, %1" : "+v"(a0) : "v"(i0)); asm volatile("v_cvt_i32_f32 %0, %1" : "+v"(i1) : "v"(a1)); asm volatile("v_cvt_f32_i32 %0, %1" : "+v"(a2) : "v"(i2)); asm volatile("v_cvt_i32_f32 %0, %1" : "+v"(i3) : "v"(a3));) }
        if (K == 26) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(db)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(db)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(db)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(db));) }
        if (K == 27) { REP16(asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(i0) : "s20"); asm volatile("v_readfirstlane_b32 s21, %0" :: "v"(i1) : "s21"); asm volatile("v_readlane_b32 s22, %0, 5" :: "v"(i2) : "s22"); asm volatile("v_readfirstlane_b32 s23, %0" :: "v"(i3) : "s23");) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(i0 + i1 + i2 + i3) + (float)(d0 + d1);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K>
void run(const char* name, int per_inst, int wg_per_cu, int n_cu, float* d_out, unsigned long long* d_cyc)
{
    const int grid = n_cu * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < grid; ++i) avg += (double)h[i];
    avg /= grid;
    const double inst_per_wave = (double)ITER * 16 * per_inst;
    // every SIMD runs wg_per_cu waves (a 256-thread workgroup puts one wave on each SIMD); in-kernel cycles per wave
    // / (instructions per wave x waves per SIMD) = SIMD cycles per wave-instruction
    (void)avg;
    printf("%-36s %d waves/SIMD: %7.1f G wave-instr/s (wall %.3f ms)\n", name, wg_per_cu, inst_per_wave * 4.0 * grid / (ms * 1e-3) / 1e9, ms);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    float* d_out; unsigned long long* d_cyc;
    hipMalloc((void**)&d_out, sizeof(float) * 256 * n_cu * 8);
    hipMalloc((void**)&d_cyc, sizeof(unsigned long long) * n_cu * 8);
    for (int w : {1, 2, 4, 8}) {
        if (argc > 1 && atoi(argv[1]) != w) continue;
        run<0>("v_add_f32", 4, w, n_cu, d_out, d_cyc);
        run<1>("v_fma_f32", 4, w, n_cu, d_out, d_cyc);
        run<2>("v_add_u32", 4, w, n_cu, d_out, d_cyc);
        run<3>("v_cmp_lt_f32 + v_cndmask_b32", 4, w, n_cu, d_out, d_cyc);
        run<4>("v_mul_lo_u32", 4, w, n_cu, d_out, d_cyc);
        run<5>("v_fma_f64", 4, w, n_cu, d_out, d_cyc);
        run<6>("v_max_f32 |x|", 4, w, n_cu, d_out, d_cyc);
        run<7>("and / lshl / xor / mov", 4, w, n_cu, d_out, d_cyc);
        run<8>("v_pk_add_f32", 4, w, n_cu, d_out, d_cyc);
        run<9>("v_add_f32 / v_add_u32 alternating", 4, w, n_cu, d_out, d_cyc);
        run<10>("v_rcp_f32", 4, w, n_cu, d_out, d_cyc);
        run<11>("v_cmp f32 / u32", 4, w, n_cu, d_out, d_cyc);
        run<12>("v_max_f32 (no modifier)", 4, w, n_cu, d_out, d_cyc);
        run<13>("v_add_f32 with |x| (VOP3)", 4, w, n_cu, d_out, d_cyc);
        run<14>("v_cndmask_b32 (vcc fixed)", 4, w, n_cu, d_out, d_cyc);
        run<15>("v_cmp_lt_f32 e64 -> sgpr pair", 4, w, n_cu, d_out, d_cyc);
        run<16>("v_sub_co_u32 + v_addc_co_u32", 4, w, n_cu, d_out, d_cyc);
        run<17>("v_mov_b32 dpp quad_perm", 4, w, n_cu, d_out, d_cyc);
        run<18>("v_bcnt_u32_b32 / v_alignbit_b32", 4, w, n_cu, d_out, d_cyc);
        run<19>("v_lshl_add_u32 / v_add3_u32", 4, w, n_cu, d_out, d_cyc);
        run<20>("v_mul_f32", 4, w, n_cu, d_out, d_cyc);
        run<21>("v_mad_u32_u24 / v_mul_u32_u24", 4, w, n_cu, d_out, d_cyc);
        run<22>("v_min_u32 / v_max_i32", 4, w, n_cu, d_out, d_cyc);
        run<23>("v_cmp_gt_i32 + v_addc_co_u32", 4, w, n_cu, d_out, d_cyc);
        run<24>("v_sub_f32 / v_mac? (v_fmac_f32)", 4, w, n_cu, d_out, d_cyc);
        run<25>("v_cvt_f32_i32 / v_cvt_i32_f32", 4, w, n_cu, d_out, d_cyc);
        run<26>("v_add_f64 / v_mul_f64", 4, w, n_cu, d_out, d_cyc);
        run<27>("v_readlane_b32 / v_readfirstlane", 4, w, n_cu, d_out, d_cyc);
    }
    return 0;
}
