// GPU-box microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the path-tracing kernels are made
// of, on gfx950, no MFMA anywhere.  For each instruction: a loop of 64 independent copies (8 register chains x 8) per iteration, timed with
// s_memtime by one wave per SIMD (1 wave: issue cost incl. dependency stalls of an 8-deep rotation) and by 4 waves per SIMD
// (throughput).    hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 2000
typedef float v2 __attribute__((ext_vector_type(2)));

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define REP8(X) X X X X X X X X

template <int KIND>
__global__ void k(unsigned long long* out, float seed) {
    v2 a[8]; float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = v2{seed + i + threadIdx.x, seed * 2 + i}; s[i] = seed + i + threadIdx.x * 0.5f; }
    v2 b = {1.0001f, 0.9999f}; float sb = 1.0001f, sc = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) {      // v_fma_f32
#define I(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[n]) : "v"(sb), "v"(sc));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 1) {   // v_pk_fma_f32
#define I(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[n]) : "v"(b));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 2) {   // v_pk_mul_f32
#define I(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[n]) : "v"(b));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 3) {   // v_pk_add_f32
#define I(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[n]) : "v"(b));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 4) {   // v_rcp_f32
#define I(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[n]));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 5) {   // v_sqrt_f32
#define I(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[n]));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 6) {   // v_mul_f32
#define I(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[n]) : "v"(sb));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 7) {   // v_cndmask_b32 (vcc)
#define I(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[n]) : "v"(sb) : );
            REP8(BODY8(I))
#undef I
        } else if (KIND == 8) {   // v_cmp_lt_f32 -> sgpr pair + v_cndmask_b32 e64 reading it (the select idiom)
#define I(n) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(s[n]) : "v"(sb) : "s20", "s21");
            REP8(BODY8(I))
#undef I
        } else if (KIND == 9) {   // v_div_fixup_f32
#define I(n) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(s[n]) : "v"(sb), "v"(sc));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 10) {  // v_div_scale_f32
#define I(n) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(s[n]) : "v"(sb) : "vcc");
            REP8(BODY8(I))
#undef I
        } else if (KIND == 11) {  // dependent chain of v_fma_f32 (one register)
#define I(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[0]) : "v"(sb), "v"(sc));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 12) {  // dependent chain of v_pk_fma_f32
#define I(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(b));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 13) {  // v_min_f32
#define I(n) asm volatile("v_min_f32 %0, %0, %1" : "+v"(s[n]) : "v"(sb));
            REP8(BODY8(I))
#undef I
        } else if (KIND == 14) {  // v_mad_u64_u32
            // skipped
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0; for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    if (threadIdx.x == 0 && acc == 12345.678f) out[1023] = 1;      // keep the results alive
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND> void run(const char* name, unsigned long long* d) {
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;      // waves per CU = 4 SIMDs x wps
        const int blocks_per_cu = (64 * 4 * wps) / threads;
        const int blocks = 256 * blocks_per_cu;
        k<KIND><<<blocks, threads>>>(d, 1.0f);      // warm
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<KIND><<<blocks, threads>>>(d, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
        const double per = mean / (ITERS * 64.0);      // s_memtime ticks (100 MHz? or shader clock) per instruction per wave
        // wall-clock based: instructions per SIMD = ITERS * 64 * wps; ns per instruction-slot
        const double ns_per = ms * 1e6 / (ITERS * 64.0 * wps);
        printf("%-22s waves/SIMD %d  memtime ticks/instr/wave %7.3f  wall ns per instr per SIMD %6.3f  (x2.4GHz = %5.2f cyc)\n", name, wps, per, ns_per, ns_per * 2.4);
    }
}

int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 65536);
    run<0>("v_fma_f32", d); run<6>("v_mul_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d);
    run<4>("v_rcp_f32", d); run<5>("v_sqrt_f32", d); run<7>("v_cndmask_b32 vcc", d); run<8>("v_cmp+nop+cndmask", d); run<9>("v_div_fixup_f32", d);
    run<10>("v_div_scale_f32", d); run<13>("v_min_f32", d); run<11>("v_fma_f32 dep chain", d); run<12>("v_pk_fma_f32 dep chain", d);
    return 0;
}
