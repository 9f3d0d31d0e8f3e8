/* Probe of gfx950's 16-byte LDS-direct load (global_load_lds_dwordx4) as inline asm: where a lane's 16 bytes land (m0 + lane * 16),
 * that masked-off lanes leave LDS alone, and that the load counts in vmcnt in issue order.
 *   hipcc -O3 --offload-arch=gfx950 tools/lds_dma_probe.hip -o variants/lds_dma_probe && variants/lds_dma_probe */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void dma16(const void *gaddr, uint32_t lds_off, unsigned long long mask)
{
    unsigned long long saved;
    uint32_t m0_saved;
    __asm__ volatile(
        "s_mov_b32 %[m0s], m0\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mask]\n\t"
        "s_mov_b32 m0, %[m0v]\n\t"
        "global_load_lds_dwordx4 %[addr], off\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_mov_b32 m0, %[m0s]"
        : [sv] "=&s"(saved), [m0s] "=&s"(m0_saved)
        : [mask] "s"(mask), [m0v] "s"(lds_off), [addr] "v"(gaddr)
        : "memory");
}
__global__ void k(const uint64_t *src, const uint64_t *far, uint64_t *dst, unsigned long long mask)
{
    extern __shared__ uint64_t lds[];
    for (int i = threadIdx.x; i < 768; i += 64) lds[i] = 0xDEADull;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t *slot = lds + 128, *slot2 = lds + 256, *slot3 = lds + 384;
    const uint32_t off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)(char *)slot);
    /* A: a far-away (cold) source, all lanes; B: one lane only; C: the masked pattern */
    dma16((const char *)far + lane * 16, off + 1024, ~0ull);
    dma16((const char *)src, off + 2048, 1ull);
    dma16((const char *)src + lane * 16, off, mask);
    __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory"); /* A must be there now, whatever B and C do */
    const uint64_t a0 = slot2[lane], a1 = slot2[64 + lane];
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dst[lane] = slot[lane];
    dst[64 + lane] = slot[64 + lane];
    dst[128 + lane] = a0;
    dst[192 + lane] = a1;
    dst[256 + lane] = slot3[lane];
    dst[320 + lane] = lds[lane];
}
int main()
{
    const size_t far_words = (size_t)1 << 27; /* 1 GB away from anything cached */
    std::vector<uint64_t> h(128);
    for (int i = 0; i < 128; i += 1) h[i] = 1000 + i;
    uint64_t *src, *far, *dst;
    hipMalloc(&src, 128 * 8); hipMalloc(&far, far_words * 8); hipMalloc(&dst, 384 * 8);
    hipMemcpy(src, h.data(), 128 * 8, hipMemcpyHostToDevice);
    std::vector<uint64_t> hf(128);
    for (int i = 0; i < 128; i += 1) hf[i] = 5000 + i;
    hipMemcpy(far + far_words - 128, hf.data(), 128 * 8, hipMemcpyHostToDevice);
    const unsigned long long mask = 0x00000000F0F0FFFFull;
    int bad = 0;
    for (int rep = 0; rep < 50; rep += 1)
    {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 768 * 8, 0, src, far + far_words - 128, dst, mask);
        std::vector<uint64_t> out(384);
        hipMemcpy(out.data(), dst, 384 * 8, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l += 1)
        {
            const bool on = (mask >> l) & 1;
            for (int w = 0; w < 2; w += 1)
            {
                const uint64_t got = out[2 * l + w], want = on ? 1000 + 2 * l + w : 0xDEADull;
                if (got != want) { if (bad < 10) printf("C lane %d word %d: %llu want %llu\n", l, w, (unsigned long long)got, (unsigned long long)want); bad += 1; }
            }
        }
        for (int i = 0; i < 128; i += 1)
            if (out[128 + i] != 5000 + (uint64_t)i) { if (bad < 10) printf("A word %d after vmcnt(2): %llu\n", i, (unsigned long long)out[128 + i]); bad += 1; }
        if (out[256] != 1000 || out[257] != 1001 || out[258] != 0xDEADull) { printf("B: %llu %llu %llu\n", (unsigned long long)out[256], (unsigned long long)out[257], (unsigned long long)out[258]); bad += 1; }
        for (int i = 0; i < 64; i += 1) if (out[320 + i] != 0xDEADull) bad += 1;
    }
    printf(bad ? "lds_dma_probe: %d mismatches\n" : "lds_dma_probe: ok (lane l -> m0 + 16 l, masked lanes untouched, vmcnt counts in issue order)\n", bad);
    return bad != 0;
}
