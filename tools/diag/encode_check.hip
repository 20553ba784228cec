// Developer check (GPU box): encode16 of kpal_device.hpp (v_dot4 gathers) against its shift-or form on random and edge bytes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/encode_check tools/diag/encode_check.hip && /tmp/encode_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../kpal_amd/csrc/kpal_device.hpp"
using namespace kpal;
__device__ Chunk encode16_shifts(uint4 v)
{
    uint32_t t0, t1, t2, t3, z0, z1, z2, z3;
    classify4(v.x, t0, z0); classify4(v.y, t1, z1); classify4(v.z, t2, z2); classify4(v.w, t3, z3);
    const uint32_t c0 = codes_to_top_byte(t0), c1 = codes_to_top_byte(t1), c2 = codes_to_top_byte(t2), c3 = codes_to_top_byte(t3);
    const uint32_t c01 = __builtin_amdgcn_perm(c0, c1, 0x07030000u), c23 = __builtin_amdgcn_perm(c2, c3, 0x07030000u);
    const uint32_t f01 = flags_to_top_byte((z0 >> 3) | (z1 >> 7)), f23 = flags_to_top_byte((z2 >> 3) | (z3 >> 7));
    Chunk r;
    r.codes = __builtin_amdgcn_perm(c01, c23, 0x07060302u);
    r.bad = __builtin_amdgcn_perm(f01, f23, 0x0c0c0703u);
    return r;
}
__global__ void both(const uint4 *in, uint4 *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Chunk a = encode16(in[i]), b = encode16_shifts(in[i]);
    out[i] = make_uint4(a.codes, a.bad, b.codes, b.bad);
}
int main()
{
    const int n = 1 << 20;
    std::vector<uint8_t> h((size_t)n * 16);
    const char *alpha = "ACGTacgtNn\n-RY";
    for (size_t i = 0; i < h.size(); ++i) h[i] = (rand() % 10) ? (uint8_t)alpha[rand() % 14] : (uint8_t)(rand() & 255);
    uint4 *din, *dout;
    hipMalloc(&din, h.size());
    hipMalloc(&dout, (size_t)n * 16);
    hipMemcpy(din, h.data(), h.size(), hipMemcpyHostToDevice);
    both<<<n / 256, 256>>>(din, dout, n);
    std::vector<uint4> o(n);
    hipMemcpy(o.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (o[i].x != o[i].z || o[i].y != o[i].w) {
            if (bad++ < 5) {
                printf("chunk %d: dot4 codes %08x bad %04x   shifts codes %08x bad %04x   bytes", i, o[i].x, o[i].y, o[i].z, o[i].w);
                for (int j = 0; j < 16; ++j) printf(" %02x", h[(size_t)i * 16 + j]);
                printf("\n");
            }
        }
    printf(bad ? "ENCODE MISMATCH in %d chunks\n" : "ENCODE_OK %d\n", bad);
    return bad != 0;
}
