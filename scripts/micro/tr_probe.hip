// dev probe: what ds_read_b64_tr_b16 returns. LDS holds lds[i] = i (16-bit); lane l reads at element address a(l); the four
// 16-bit results of every lane are printed as source element indices.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int a;
    if (mode == 0) a = l * 4;                              // lane-linear 8 bytes
    else if (mode == 1) a = (l & 15) * 64 + (l >> 4) * 4;  // 16 rows of 64 elements, lane group g reads columns 4g..4g+3
    else a = (l & 15) * 32 + (l >> 4) * 4;                 // rows of 32 elements (64 B, a V row of head dim 32)
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + a));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 512);
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 64>>>(d, mode);
        short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
