// Hardware probe: lane / element mapping of ds_read_b64_tr_b16 (gfx950 LDS transpose read).
// LDS holds u16 value = row * 256 + col of a [64][64] matrix (row stride 64 halfs).  Within each 16-lane group lane t supplies
// the address of 4 contiguous halfs: row (t >> 2), columns 4 * (t & 3) .. +3 of the group's 4 x 16 block; the probe prints
// what every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ unsigned short s[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) s[i] = (unsigned short)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, g = l >> 4;
    // group g: rows 4 * (g >> 1) .. +3, columns 16 * (g & 1) .. +15
    const int row = 4 * (g >> 1) + (t >> 2), col = 16 * (g & 1) + 4 * (t & 3);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(s + row * 64 + col));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] >> 8, h[l * 4 + j] & 255); printf("\n"); }
    return 0;
}
