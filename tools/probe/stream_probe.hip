// Traffic-only stand-in for k_step_fused<4,1,256>: the same per-house byte streams (53 B read, 46 B written per house:
// 13 four-byte + 1 one-byte read streams, 3 four-byte + 2 one-byte temporal write streams, 8 four-byte non-temporal write
// streams), trivial arithmetic, and the memory LAYOUT as the experiment variable.  What rate does the access pattern itself
// reach on this machine, and does placing a workgroup's streams next to each other beat 27 arrays far apart?
//
//   stream_probe <tiles> <layout> [iters] [mode: 0 full | 1 reads only | 2 writes only]
//      layout: 0 = separate arrays, staggered by 2304 B (the product's slab layout)
//                                                     1 = tile-struct: all 27 streams of a 1024-house tile contiguous
//                                                     2 = class-struct: [tile][13 reads] , [tile][3+2 state writes share the reads' place], [tile][8 outputs]
//                                                     3 = separate arrays, no stagger
// One workgroup of 256 threads per 1024-house tile (= one env of the C3 shape); prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NR4 = 13, NW4 = 3, NWNT = 8;   // four-byte streams: read (3 of them are also written), temporal write, non-temporal write
constexpr int TILE = 1024, THREADS = 256;

struct Streams {
  const char* r4[NR4]; size_t r4s[NR4];
  const char* r1;      size_t r1s;
  char* w4[NW4];       size_t w4s[NW4];
  char* w1[2];         size_t w1s[2];
  char* wnt[NWNT];     size_t wnts[NWNT];
};

typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: the full pattern; 1: the 53 B/house of reads only (one float per workgroup written if the sum is NaN: never);
// 2: the 46 B/house of writes only
template <int MODE>
__global__ __launch_bounds__(THREADS) void k_probe(Streams s) {
  const size_t e = blockIdx.x;
  const int t = threadIdx.x;
  float4 acc = make_float4((float)t, 1.f, 2.f, 3.f);
  uchar4 f = make_uchar4(1, 0, 1, 0);
  if (MODE != 2) {
    float4 v[NR4];
#pragma unroll
    for (int i = 0; i < NR4; ++i) v[i] = *reinterpret_cast<const float4*>(s.r4[i] + e * s.r4s[i] + (size_t)t * 16);
    f = *reinterpret_cast<const uchar4*>(s.r1 + e * s.r1s + (size_t)t * 4);
#pragma unroll
    for (int i = 0; i < NR4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
  }
  if (MODE == 1) {
    if (acc.x + acc.y + acc.z + acc.w + (float)f.x == 12345.678f) s.w4[0][e * s.w4s[0]] = 1;
    return;
  }
#pragma unroll
  for (int i = 0; i < NW4; ++i) {
    float4 o = make_float4(acc.x + i, acc.y * 0.5f, acc.z, acc.w);
    *reinterpret_cast<float4*>(s.w4[i] + e * s.w4s[i] + (size_t)t * 16) = o;
  }
  const uchar4 g = make_uchar4(f.x ^ 1, f.y ^ (acc.x > 0.f), f.z, f.w);
  *reinterpret_cast<uchar4*>(s.w1[0] + e * s.w1s[0] + (size_t)t * 4) = g;
  *reinterpret_cast<uchar4*>(s.w1[1] + e * s.w1s[1] + (size_t)t * 4) = f;
#pragma unroll
  for (int i = 0; i < NWNT; ++i) {
    v4f o = {acc.x * (i + 1), acc.y, acc.z, acc.w};
    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(s.wnt[i] + e * s.wnts[i] + (size_t)t * 16));
  }
}

int main(int argc, char** argv) {
  const size_t tiles = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4096;
  const int layout = argc > 2 ? atoi(argv[2]) : 0;
  const int iters = argc > 3 ? atoi(argv[3]) : 100;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const size_t B4 = (size_t)TILE * 4, B1 = TILE;   // bytes per tile of a four-byte / one-byte stream
  // logical streams: 13 R4 (0..2 = state, also written), 1 R1 (flags, also written), 1 W1 (actions), 8 WNT
  Streams s{};
  size_t total = 0;
  char* base = nullptr;
  auto align = [](size_t x, size_t a) { return (x + a - 1) / a * a; };
  if (layout == 0 || layout == 3) {
    std::vector<size_t> off;
    size_t o = 0;
    const int n = NR4 + 1 + 1 + NWNT;   // arrays: 13 + flags + actions + 8
    for (int i = 0; i < n; ++i) {
      o = align(o, 256) + (layout == 0 ? 2304 * (size_t)(i % 16) : 0);
      o = align(o, 256);
      off.push_back(o);
      o += tiles * ((i == NR4 || i == NR4 + 1) ? B1 : B4);
    }
    total = align(o, 256);
    CK(hipMalloc(&base, total));
    for (int i = 0; i < NR4; ++i) { s.r4[i] = base + off[i]; s.r4s[i] = B4; }
    s.r1 = base + off[NR4]; s.r1s = B1;
    for (int i = 0; i < NW4; ++i) { s.w4[i] = base + off[i]; s.w4s[i] = B4; }
    s.w1[0] = base + off[NR4]; s.w1s[0] = B1;
    s.w1[1] = base + off[NR4 + 1]; s.w1s[1] = B1;
    for (int i = 0; i < NWNT; ++i) { s.wnt[i] = base + off[NR4 + 2 + i]; s.wnts[i] = B4; }
  } else if (layout == 1) {
    const size_t st = NR4 * B4 + 2 * B1 + NWNT * B4;   // 13*4K + 2K + 32K = 86 KiB per tile
    total = tiles * st;
    CK(hipMalloc(&base, total));
    size_t o = 0;
    for (int i = 0; i < NR4; ++i) { s.r4[i] = base + o; s.r4s[i] = st; if (i < NW4) { s.w4[i] = base + o; s.w4s[i] = st; } o += B4; }
    s.r1 = base + o; s.r1s = st; s.w1[0] = base + o; s.w1s[0] = st; o += B1;
    s.w1[1] = base + o; s.w1s[1] = st; o += B1;
    for (int i = 0; i < NWNT; ++i) { s.wnt[i] = base + o; s.wnts[i] = st; o += B4; }
  } else {
    const size_t st_in = NR4 * B4 + 2 * B1, st_out = NWNT * B4;
    total = tiles * (st_in + st_out) + 4096;
    CK(hipMalloc(&base, total));
    char* outb = base + align(tiles * st_in, 256) + 2304;
    size_t o = 0;
    for (int i = 0; i < NR4; ++i) { s.r4[i] = base + o; s.r4s[i] = st_in; if (i < NW4) { s.w4[i] = base + o; s.w4s[i] = st_in; } o += B4; }
    s.r1 = base + o; s.r1s = st_in; s.w1[0] = base + o; s.w1s[0] = st_in; o += B1;
    s.w1[1] = base + o; s.w1s[1] = st_in;
    o = 0;
    for (int i = 0; i < NWNT; ++i) { s.wnt[i] = outb + o; s.wnts[i] = st_out; o += B4; }
  }
  CK(hipMemset(base, 0, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3((unsigned)tiles), dim3(THREADS), 0, 0, s);
    else if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3((unsigned)tiles), dim3(THREADS), 0, 0, s);
    else hipLaunchKernelGGL(k_probe<0>, dim3((unsigned)tiles), dim3(THREADS), 0, 0, s);
  };
  for (int i = 0; i < 10; ++i) launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = best / iters * 1e3, houses = (double)tiles * TILE;
  const double bytes = mode == 1 ? 53.0 : (mode == 2 ? 46.0 : 99.0);
  printf("{\"probe\": \"stream\", \"mode\": \"%s\", \"layout\": %d, \"tiles\": %zu, \"houses\": %.0f, \"us\": %.2f, \"bytes_per_house\": %.0f, \"GBps\": %.1f, \"alloc_MB\": %.1f}\n",
         mode == 1 ? "reads" : (mode == 2 ? "writes" : "full"), layout, tiles, houses, us, bytes, houses * bytes / us * 1e-3, total / 1e6);
  CK(hipFree(base));
  return 0;
}
