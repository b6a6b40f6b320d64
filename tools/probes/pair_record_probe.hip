#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float agent_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swap_neighbour(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float swap_neighbour(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ void load_agent_two(const float4* p, const float4* q, float4& a, float4& b) {
    agent_f4 x, y;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(p), "v"(q) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w);
}
__device__ __forceinline__ void store_agent_f4(float4* p, float4 v) {
    agent_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4* rec(float4* t, int body) { return t + (size_t)body * 2; }
__global__ void release(float4* table, const int* body, const int* publish, const float* lin, const float* ang, const unsigned* number) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const bool pub = publish[t] != 0;
    if (__builtin_amdgcn_ballot_w64(pub) == 0) return;
    const bool odd = (threadIdx.x & 1u) != 0u;
    const int mine = pub ? body[t] : -1, theirs = swap_neighbour(mine);
    const float n = __uint_as_float(number[t] + 1u), their_n = swap_neighbour(n);
    const float lx = lin[3 * t], ly = lin[3 * t + 1], lz = lin[3 * t + 2], ax = ang[3 * t], ay = ang[3 * t + 1], az = ang[3 * t + 2];
    const float gx = swap_neighbour(odd ? lx : ax), gy = swap_neighbour(odd ? ly : ay), gz = swap_neighbour(odd ? lz : az);
    const int first_body = odd ? theirs : mine, second_body = odd ? mine : theirs;
    const float4 first = odd ? make_float4(gx, gy, gz, their_n) : make_float4(lx, ly, lz, n);
    const float4 second = odd ? make_float4(ax, ay, az, n) : make_float4(gx, gy, gz, their_n);
    if (first_body >= 0) store_agent_f4(rec(table, first_body) + (odd ? 1 : 0), first);
    if (second_body >= 0) store_agent_f4(rec(table, second_body) + (odd ? 1 : 0), second);
}
__global__ void acquire(float4* table, const int* body, const int* need, float* out) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    bool need_a = need[t] != 0;
    const bool odd = (threadIdx.x & 1u) != 0u;
    float4 l = make_float4(-1, -1, -1, -1), w = l;
    const int b = need_a ? body[t] : 0, theirs = swap_neighbour(b);
    float4* a1 = rec(table, odd ? theirs : b) + (odd ? 1 : 0);
    float4* a2 = rec(table, odd ? b : theirs) + (odd ? 1 : 0);
    float4 x1, x2;
    load_agent_two(a1, a2, x1, x2);
    const float4 send = odd ? x1 : x2;
    const float4 got = make_float4(swap_neighbour(send.x), swap_neighbour(send.y), swap_neighbour(send.z), swap_neighbour(send.w));
    l = odd ? got : x1; w = odd ? x2 : got;
    if (!need_a) { l = make_float4(-1, -1, -1, -1); w = l; }
    float* o = out + 8 * t;
    o[0] = l.x; o[1] = l.y; o[2] = l.z; o[3] = l.w; o[4] = w.x; o[5] = w.y; o[6] = w.z; o[7] = w.w;
}
int main() {
    const int n = 512, bodies = 4096;
    std::vector<int> body(n), pub(n); std::vector<float> lin(3 * n), ang(3 * n); std::vector<unsigned> num(n);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<int> perm(bodies); for (int i = 0; i < bodies; ++i) perm[i] = i;
    for (int i = bodies - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
    for (int t = 0; t < n; ++t) {
        body[t] = perm[t]; pub[t] = (t / 64 == 3) ? 0 : (t / 64 == 4 ? 1 : (rnd() % 3 != 0)); num[t] = 1000 + t;
        for (int k = 0; k < 3; ++k) { lin[3 * t + k] = t + 0.1f * (k + 1); ang[3 * t + k] = -(t + 0.1f * (k + 1)); }
    }
    float4* table; int *d_body, *d_pub; float *d_lin, *d_ang, *d_out; unsigned* d_num;
    CHECK(hipMalloc(&table, bodies * 32)); CHECK(hipMemset(table, 0, bodies * 32));
    CHECK(hipMalloc(&d_body, n * 4)); CHECK(hipMalloc(&d_pub, n * 4)); CHECK(hipMalloc(&d_lin, n * 12)); CHECK(hipMalloc(&d_ang, n * 12)); CHECK(hipMalloc(&d_num, n * 4)); CHECK(hipMalloc(&d_out, n * 32));
    CHECK(hipMemcpy(d_body, body.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_pub, pub.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lin, lin.data(), n * 12, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ang, ang.data(), n * 12, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_num, num.data(), n * 4, hipMemcpyHostToDevice));
    release<<<n / 256, 256>>>(table, d_body, d_pub, d_lin, d_ang, d_num);
    CHECK(hipDeviceSynchronize());
    std::vector<float> h(bodies * 8), expect(bodies * 8, 0.0f);
    CHECK(hipMemcpy(h.data(), table, bodies * 32, hipMemcpyDeviceToHost));
    for (int t = 0; t < n; ++t) if (pub[t]) {
        float* e = &expect[body[t] * 8]; unsigned nn = num[t] + 1; float nf; memcpy(&nf, &nn, 4);
        e[0] = lin[3 * t]; e[1] = lin[3 * t + 1]; e[2] = lin[3 * t + 2]; e[3] = nf; e[4] = ang[3 * t]; e[5] = ang[3 * t + 1]; e[6] = ang[3 * t + 2]; e[7] = nf;
    }
    int bad = 0;
    for (int i = 0; i < bodies * 8; ++i) bad += memcmp(&h[i], &expect[i], 4) != 0;
    printf("release: %d wrong words of %d\n", bad, bodies * 8);
    // acquire: table = expect (complete records for every body), random needs
    for (int b = 0; b < bodies; ++b) for (int k = 0; k < 8; ++k) expect[b * 8 + k] = b * 10.0f + k;
    CHECK(hipMemcpy(table, expect.data(), bodies * 32, hipMemcpyHostToDevice));
    std::vector<int> need(n); for (int t = 0; t < n; ++t) need[t] = (t / 64 == 2) ? 0 : (t / 64 == 5 ? 1 : (rnd() % 3 != 0));
    CHECK(hipMemcpy(d_pub, need.data(), n * 4, hipMemcpyHostToDevice));
    acquire<<<n / 256, 256>>>(table, d_body, d_pub, d_out);
    CHECK(hipDeviceSynchronize());
    std::vector<float> out(n * 8); CHECK(hipMemcpy(out.data(), d_out, n * 32, hipMemcpyDeviceToHost));
    bad = 0;
    for (int t = 0; t < n; ++t) for (int k = 0; k < 8; ++k) { float e = need[t] ? body[t] * 10.0f + k : -1.0f; bad += out[t * 8 + k] != e; }
    printf("acquire: %d wrong words of %d\n", bad, n * 8);
    return 0;
}
