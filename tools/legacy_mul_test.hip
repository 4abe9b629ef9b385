#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const float *a, const float *b, float *o1, float *o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { o1[i] = a[i] * b[i]; float r; asm volatile("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b[i])); o2[i] = r; }
}
int main() {
    const int n = 1 << 24;
    std::vector<float> a(n), b(n), o1(n), o2(n);
    std::mt19937 g(1);
    const uint32_t special[] = {0, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x7fa00001u, 0xffc12345u, 1u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0x3f800000u, 0xbf800000u};
    for (int i = 0; i < n; i++) {
        uint32_t x = g(), y = g();
        if (i < 14 * 14) { x = special[i / 14]; y = special[i % 14]; }
        else if (i < 200000) { x = special[i % 14]; }           // special x random
        else if (i < 400000) { y = special[i % 14]; }
        else if (i < 4000000) { x = (x & 0x807fffffu) | ((uint32_t)(100 + (x >> 24) % 60) << 23); y = (y & 0x807fffffu) | ((uint32_t)(100 + (y >> 24) % 60) << 23); }   // ordinary magnitudes
        else if (i < 6000000) { x = (x & 0x807fffffu) | ((uint32_t)((x >> 24) % 30) << 23); y = (y & 0x807fffffu) | ((uint32_t)(110 + (y >> 24) % 30) << 23); }    // products near the denormal range
        memcpy(&a[i], &x, 4); memcpy(&b[i], &y, 4);
    }
    float *da, *db, *d1, *d2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, d1, d2, n);
    hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    long diff_zero = 0, diff_other = 0; int shown = 0;
    for (int i = 0; i < n; i++) {
        uint32_t u1, u2, x, y; memcpy(&u1, &o1[i], 4); memcpy(&u2, &o2[i], 4); memcpy(&x, &a[i], 4); memcpy(&y, &b[i], 4);
        if (u1 == u2) continue;
        const bool zero_operand = (x << 1) == 0 || (y << 1) == 0;
        if (zero_operand) diff_zero++; else { diff_other++; if (shown++ < 10) printf("diff: %08x * %08x -> ieee %08x legacy %08x\n", x, y, u1, u2); }
    }
    printf("differences with a zero operand: %ld, without: %ld (of %d)\n", diff_zero, diff_other, n);
    // what legacy gives for zero x special
    for (int i = 0; i < 14; i++) { uint32_t u1, u2; memcpy(&u1, &o1[i], 4); memcpy(&u2, &o2[i], 4); printf("0 * %08x: ieee %08x legacy %08x\n", special[i], u1, u2); }
    for (int i = 14; i < 28; i++) { uint32_t u1, u2; memcpy(&u1, &o1[i], 4); memcpy(&u2, &o2[i], 4); printf("-0 * %08x: ieee %08x legacy %08x\n", special[i - 14], u1, u2); }
    return 0;
}
