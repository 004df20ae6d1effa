// Hand-over of one 32 KB tile (64 x 64 fp64) between two workgroups INSIDE one launch, by publication form and by placement
// (same XCD / different XCDs): the go / no-go measurement of VERDICT r4 item 2 (a tile Cholesky whose thin levels hand their tiles
// over through flags instead of launch boundaries).  Dev tool; build: hipcc -O3 --offload-arch=gfx950 -o xcd_handover xcd_handover.hip
//
// Two 512-thread workgroups (the potrf's size; 96 KB of LDS each so that they cannot share a compute unit) play ping-pong for R rounds:
// ping writes tile 0 (values depend on round and element), publishes; pong waits, reads and CHECKS every word, writes tile 1, publishes;
// ping waits, reads and checks.  Time per hand-over = ping's wall clock / 2R.  Forms:
//   0 fence   plain 16-B stores -> barrier -> lane 0: release fence (buffer_wbl2 sc1) + vmcnt(0) -> relaxed agent flag;
//             consumer: lane 0 polls -> acquire fence (buffer_inv sc1) -> barrier -> plain loads               [valid on any placement]
//   1 sc1     16-B sc1 (write-through) stores -> every wave vmcnt(0) -> barrier -> flag; consumer: poll -> barrier -> sc1 loads [valid anywhere]
//   2 l2      plain 16-B stores -> every wave vmcnt(0) -> barrier -> flag; consumer: poll -> barrier -> sc1 loads (L1 bypassed, served by
//             the XCD's L2: sees the producer's lines only if both sit on the SAME XCD)                       [same XCD only]
//   3 l2+inv  as 2, consumer: poll -> buffer_inv sc1 -> barrier -> plain loads                                [same XCD only]
// and the same with an empty tile (flag alone).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ inline int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }
__device__ inline int hw_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }

__device__ inline void store16(d2* p, d2 v, bool sc1) {
  if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
}
__device__ inline void load4_sc1(const d2* p0, const d2* p1, const d2* p2, const d2* p3, d2& a, d2& b, d2& c, d2& d) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n global_load_dwordx4 %1, %5, off sc1\n global_load_dwordx4 %2, %6, off sc1\n global_load_dwordx4 %3, %7, off sc1\n s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
__device__ inline double value_of(int round, int who, int e) { return (double)(round * 2 + who) * 4096.0 + (double)e; }

struct Args {
  double* tiles;            // 2 tiles of 4096 doubles
  unsigned long long* flags;   // [0] ping -> pong, [16] pong -> ping (different lines)
  int* info;                // [0..1] xcc of ping / pong, [2..3] hw id, [4] mismatching words, [5] time-outs
  long long* ticks;         // wall clock ticks of ping (100 MHz)
  int ping, pong, form, rounds, tile16;   // tile16: 16-B pieces per thread (4 = a 32 KB tile, 0 = flag only)
};

__device__ inline void publish(const Args& a, int who, int round) {
  const int t = threadIdx.x;
  d2* tile = reinterpret_cast<d2*>(a.tiles + (size_t)who * 4096);
  const bool sc1 = a.form == 1;
  for (int i = 0; i < a.tile16; ++i) {
    const int q = i * 512 + t;   // 16-B piece
    d2 v; v.x = value_of(round, who, 2 * q); v.y = value_of(round, who, 2 * q + 1);
    store16(tile + q, v, sc1);
  }
  if (a.form != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    if (a.form == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __hip_atomic_store(a.flags + 16 * who, (unsigned long long)round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ inline void consume(const Args& a, int from, int round, int* bad, int* timeouts) {
  const int t = threadIdx.x;
  if (t == 0) {
    long long spins = 0;
    while (__hip_atomic_load(a.flags + 16 * from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)round) {
      if (++spins > (1ll << 22)) { ++*timeouts; break; }
    }
    if (a.form == 0 || a.form == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  const d2* tile = reinterpret_cast<const d2*>(a.tiles + (size_t)from * 4096);
  if (a.tile16 == 4) {
    d2 v[4];
    if (a.form == 1 || a.form == 2) load4_sc1(tile + t, tile + 512 + t, tile + 1024 + t, tile + 1536 + t, v[0], v[1], v[2], v[3]);
    else for (int i = 0; i < 4; ++i) v[i] = tile[i * 512 + t];
    for (int i = 0; i < 4; ++i) {
      const int q = i * 512 + t;
      if (v[i].x != value_of(round, from, 2 * q)) ++*bad;
      if (v[i].y != value_of(round, from, 2 * q + 1)) ++*bad;
    }
  }
}

__global__ void __launch_bounds__(512) k_pingpong(Args a) {
  __shared__ double pad[12288];   // 96 KB: one workgroup per compute unit
  const int b = blockIdx.x, t = threadIdx.x;
  if (b != a.ping && b != a.pong) return;
  pad[t] = 0.0;
  const int who = b == a.ping ? 0 : 1;
  if (t == 0) { a.info[who] = xcc_id(); a.info[2 + who] = hw_id(); }
  int bad = 0, timeouts = 0;
  const long long t0 = wall_clock64();
  for (int r = 1; r <= a.rounds; ++r) {
    if (who == 0) { publish(a, 0, r); consume(a, 1, r, &bad, &timeouts); }
    else { consume(a, 0, r, &bad, &timeouts); publish(a, 1, r); }
    if (__syncthreads_or(timeouts)) break;   // a flag that never arrives: give up (the other side times out as well)
  }
  const long long t1 = wall_clock64();
  if (who == 0 && t == 0) *a.ticks = t1 - t0;
  if (bad) atomicAdd(a.info + 4, bad);
  if (timeouts) atomicAdd(a.info + 5, timeouts);
  if (pad[t] == 123.0) a.tiles[0] = 1.0;
}

// the same 2R hand-overs as 2R dependent launches of a kernel that reads the tile its predecessor wrote and writes the other one (the launch boundary
// the flags would replace), timed with events
__global__ void __launch_bounds__(512) k_step(double* tiles, int round, int who, int* bad) {
  __shared__ double pad[12288];
  const int t = threadIdx.x;
  pad[t] = 0.0;
  const d2* src = reinterpret_cast<const d2*>(tiles + (size_t)(1 - who) * 4096);
  d2* dst = reinterpret_cast<d2*>(tiles + (size_t)who * 4096);
  int nb = 0;
  for (int i = 0; i < 4; ++i) {
    const int q = i * 512 + t;
    const d2 v = src[q];
    if (round > 0 && v.x != value_of(who == 0 ? round - 1 : round, 1 - who, 2 * q)) ++nb;
    d2 w; w.x = value_of(round, who, 2 * q); w.y = value_of(round, who, 2 * q + 1);
    dst[q] = w;
  }
  if (nb) atomicAdd(bad, nb);
  if (pad[t] == 123.0) tiles[0] = 1.0;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 2000;
  double* tiles; unsigned long long* flags; int* info; long long* ticks;
  hipMalloc(&tiles, 2 * 4096 * 8); hipMalloc(&flags, 64 * 8); hipMalloc(&info, 64); hipMalloc(&ticks, 8);
  const char* names[4] = {"fence (wbl2 + inv)", "sc1 stores + sc1 loads", "plain stores + sc1 loads", "plain stores + inv"};
  struct Pair { int a, b; const char* what; } pairs[] = {{0, 8, "same XCD (blocks 0, 8)"}, {0, 16, "same XCD (blocks 0, 16)"}, {0, 1, "two XCDs (blocks 0, 1)"}, {0, 4, "two XCDs (blocks 0, 4)"}};
  std::printf("# %d rounds (2 hand-overs each); ticks of the 100 MHz wall clock\n", rounds);
  for (const Pair& p : pairs)
    for (int tile16 : {4, 0})
      for (int form = 0; form < 4; ++form) {
        if (tile16 == 0 && form >= 2) continue;
        double best = 1e30; int hi[6] = {};
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(flags, 0, 64 * 8); hipMemset(info, 0, 64); hipMemset(tiles, 0, 2 * 4096 * 8);
          Args a{tiles, flags, info, ticks, p.a, p.b, form, rounds, tile16};
          hipLaunchKernelGGL(k_pingpong, dim3(64), dim3(512), 0, 0, a);
          if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
          long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(hi, info, sizeof(hi), hipMemcpyDeviceToHost);
          const double us = (double)tk / 100.0 / (2.0 * rounds);
          if (us < best) best = us;
        }
        std::printf("%-26s %-8s %-26s xcc %d -> %d  cu %03x -> %03x : %6.2f us per hand-over, %d stale words, %d time-outs\n", p.what, tile16 ? "32 KB" : "flag", names[form], hi[0], hi[1],
                    (hi[2] >> 8) & 0xff, (hi[3] >> 8) & 0xff, best, hi[4], hi[5]);
      }
  {   // the launch boundary: 2R dependent one-workgroup launches
    hipMemset(info, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(k_step, dim3(1), dim3(512), 0, 0, tiles, r, 0, info);
        hipLaunchKernelGGL(k_step, dim3(1), dim3(512), 0, 0, tiles, r, 1, info);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int bad; hipMemcpy(&bad, info, 4, hipMemcpyDeviceToHost);
      std::printf("launch boundary, one 512-thread workgroup reading + writing a 32 KB tile per launch: %6.2f us per launch (%d bad words)\n", ms * 1e3 / (2.0 * rounds), bad);
    }
  }
  return 0;
}
