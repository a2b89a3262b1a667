// Calibration of the ATTAINABLE fp16 MFMA rate on this board: the f16x2 contraction's inner
// loop in isolation -- per 16-deep step and wave 12 x v_mfma_f32_32x32x16_f16 on a 2 x 2 grid of 32 x 32
// accumulators (3 split products) -- with its operands fed three ways:
//     regs     a ring of RING pre-loaded operand sets in registers (no memory instruction in the loop)
//     +lds     the A fragments re-read from a 64 KB LDS panel every step (4 x ds_read_b128)
//     +l2      the W fragments streamed from an L2-resident 1 MB buffer every step (4 x global_load_dwordx4)
//     +both    the product kernels' loop: A from LDS, W from L2
// each with CONSTANT operands (what round 1's micro-benchmark used) and with RANDOM fp16 operands (normal,
// sigma 1: the matrix pipe's switching power depends on the data), back to back for ~1.5 s while a thread samples
// the board's hwmon power / shader-clock sensors every 10 ms.  Prints one markdown table.
// Build: hipcc --offload-arch=gfx950 -O3 -pthread mfma_ceiling.hip -o mfma_ceiling
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <random>
#include <string>
#include <thread>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int RING, bool LDSR, bool GLD>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ w, const uint4* __restrict__ a, float* out,
                                            int steps, int wstride) {
  extern __shared__ uint4 lds[];   // 64 KB "A panel": 16 steps x 4 fragments x 64 lanes x 16 B
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a[i];
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // every workgroup walks the SAME 1 MB of W (4 KB per step per wave, offset by wave): L2-resident
  const uint4* wl = w + (size_t)wave * wstride + lane;
  uint4 ring[RING][4], aring[RING][4];
  for (int d = 0; d < RING; ++d)
    for (int q = 0; q < 4; ++q) {
      ring[d][q] = wl[(d * 4 + q) * 64];
      aring[d][q] = lds[d * 256 + lane + 64 * q];
    }
  for (int s0 = 0; s0 < steps; s0 += RING) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      uint4 af[4], bf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        af[q] = aring[d][q];
        if (LDSR) aring[d][q] = lds[((s0 + d + RING) & 15) * 256 + lane + 64 * q];
        bf[q] = ring[d][q];
        if (GLD) ring[d][q] = wl[(((s0 + d + RING) & 255) * 4 + q) * 64];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int pa = t == 1 ? 1 : 0, pb = t == 0 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, af[2 * i + pa]), __builtin_bit_cast(f16x8, bf[2 * j + pb]),
                acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The single-plane bf16 form of the chains (configs[4]): ONE product per step, a wave owns FOUR 32-row blocks x 64
// channels (8 x v_mfma_f32_32x32x16_bf16 per step: 4 A fragments from LDS, 2 W fragments from L2).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int RING, bool LDSR, bool GLD>
__global__ __launch_bounds__(256, 2) void kb(const uint4* __restrict__ w, const uint4* __restrict__ a, float* out,
                                             int steps, int wstride) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a[i];
  __syncthreads();
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const uint4* wl = w + (size_t)wave * wstride + lane;
  uint4 ring[RING][2], aring[RING][4];
  for (int d = 0; d < RING; ++d) {
    for (int q = 0; q < 2; ++q) ring[d][q] = wl[(d * 2 + q) * 64];
    for (int q = 0; q < 4; ++q) aring[d][q] = lds[d * 256 + lane + 64 * q];
  }
  for (int s0 = 0; s0 < steps; s0 += RING) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      uint4 af[4], bf[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        af[q] = aring[d][q];
        if (LDSR) aring[d][q] = lds[((s0 + d + RING) & 15) * 256 + lane + 64 * q];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf[q] = ring[d][q];
        if (GLD) ring[d][q] = wl[(((s0 + d + RING) & 511) * 2 + q) * 64];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[j]),
                                                              __builtin_bit_cast(bf16x8, bf[i]), acc[i][j], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- sensors (hwmon of the GPU in use)
static std::string read_file(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return "";
  char buf[256];
  size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  return buf;
}

struct Sensors {
  std::string power, freq, cap, where;
  bool ok() const { return !power.empty(); }
};

static Sensors find_sensors(int pci_bus) {
  Sensors best;
  DIR* d = opendir("/sys/class/drm");
  if (!d) return best;
  std::vector<std::string> cards;
  while (dirent* e = readdir(d))
    if (!strncmp(e->d_name, "card", 4) && !strchr(e->d_name, '-')) cards.push_back(e->d_name);
  closedir(d);
  for (const auto& c : cards) {
    const std::string dev = "/sys/class/drm/" + c + "/device";
    char real[4096];
    if (!realpath(dev.c_str(), real)) continue;
    const char* base = strrchr(real, '/');   // 0000:bb:dd.f
    int bus = -1;
    if (base) {
      unsigned dom, b, dd, fn;
      if (sscanf(base + 1, "%x:%x:%x.%x", &dom, &b, &dd, &fn) == 4) bus = (int)b;
    }
    DIR* h = opendir((dev + "/hwmon").c_str());
    if (!h) continue;
    while (dirent* e = readdir(h)) {
      if (strncmp(e->d_name, "hwmon", 5)) continue;
      const std::string hw = dev + "/hwmon/" + e->d_name;
      Sensors s;
      for (const char* n : {"/power1_average", "/power1_input"})
        if (s.power.empty() && !read_file(hw + n).empty()) s.power = hw + n;
      if (s.power.empty()) continue;
      if (!read_file(hw + "/freq1_input").empty()) s.freq = hw + "/freq1_input";
      if (!read_file(hw + "/power1_cap").empty()) s.cap = hw + "/power1_cap";
      s.where = std::string(base ? base + 1 : "?");
      if (bus == pci_bus) {
        closedir(h);
        return s;
      }
      if (!best.ok()) best = s;
    }
    closedir(h);
  }
  return best;
}

struct Stats {
  double w_mean = 0, w_max = 0, mhz_mean = 0, mhz_min = 1e9;
  int n = 0;
};

struct Sampler {
  const Sensors& s;
  std::atomic<bool> stop{false};
  std::vector<double> w, f;
  std::thread th;
  explicit Sampler(const Sensors& s_) : s(s_) {
    th = std::thread([this] {
      while (!stop.load()) {
        if (s.ok()) {
          const std::string p = read_file(s.power);
          if (!p.empty()) w.push_back(atof(p.c_str()) / 1e6);
          if (!s.freq.empty()) {
            const std::string q = read_file(s.freq);
            if (!q.empty()) f.push_back(atof(q.c_str()) / 1e6);
          }
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
    });
  }
  Stats finish() {
    stop.store(true);
    th.join();
    Stats r;
    r.n = (int)w.size();
    // drop the first fifth: the sensors average over a window that still holds the previous state
    const size_t w0 = w.size() / 5, f0 = f.size() / 5;
    for (size_t i = w0; i < w.size(); ++i) {
      r.w_mean += w[i] / (w.size() - w0);
      if (w[i] > r.w_max) r.w_max = w[i];
    }
    for (size_t i = f0; i < f.size(); ++i) {
      r.mhz_mean += f[i] / (f.size() - f0);
      if (f[i] < r.mhz_min) r.mhz_min = f[i];
    }
    if (f.empty()) r.mhz_min = 0;
    return r;
  }
};

static uint16_t f2h(float x) {
  _Float16 h = (_Float16)x;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

struct Bufs {
  uint4 *w, *a;
  float* out;
  int wstride;
};

typedef void (*kern_t)(const uint4*, const uint4*, float*, int, int);
template <int RING, bool LDSR, bool GLD, bool BF16 = false>
static void run(const char* feed, const char* data, const Bufs& b, const Sensors& sens, double seconds, int blocks) {
  kern_t kf = BF16 ? (kern_t)&kb<RING, LDSR, GLD> : (kern_t)&k<RING, LDSR, GLD>;
  const double mfma_per_step = BF16 ? 8.0 : 12.0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int steps = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kf<<<blocks, 256, 65536>>>(b.w, b.a, b.out, 64, b.wstride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) kf<<<blocks, 256, 65536>>>(b.w, b.a, b.out, steps, b.wstride);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int n = (int)(seconds * 1e3 / (ms / 3)) + 1;
  Sampler smp(sens);
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) kf<<<blocks, 256, 65536>>>(b.w, b.a, b.out, steps, b.wstride);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  Stats st = smp.finish();
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)n * steps * mfma_per_step * blocks * 4 * 32768.0;
  const double tf = flops / ms / 1e9;
  // an MFMA of 32 x 32 x 16 occupies the SIMD's matrix pipe for 8 passes x 4 cycles: rate in MFMA-busy cycles per
  // second and SIMD, i.e. the clock the pipe would need if it never idled
  const double busy_mhz = (double)n * steps * mfma_per_step * blocks * 4 * 32.0 / 1024.0 / (ms * 1e-3) / 1e6;
  printf("| %s | %s | %.0f | %.3f | %.0f | %.0f | %.0f | %.0f | %.0f | %d |\n", feed, data, tf, tf / 2500.0,
         st.w_mean, st.w_max, st.mhz_mean, st.mhz_min, busy_mhz, st.n);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
  int dev = 0;
  hipSetDevice(dev);
  char busid[64] = {0};
  hipDeviceGetPCIBusId(busid, sizeof(busid), dev);
  unsigned dom = 0, bus = 0, dd = 0, fn = 0;
  sscanf(busid, "%x:%x:%x.%x", &dom, &bus, &dd, &fn);
  const Sensors sens = find_sensors((int)bus);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  printf("device: %s (%s, %d CUs), sensors: %s%s", prop.name, busid, prop.multiProcessorCount,
         sens.ok() ? sens.power.c_str() : "NONE", sens.ok() ? (" [" + sens.where + "]").c_str() : "");
  if (!sens.cap.empty()) printf(", power cap %.0f W", atof(read_file(sens.cap).c_str()) / 1e6);
  printf("\n\n");
  Bufs b[2];
  const int wstride = 1024 * 64 + 64;   // uint4 per wave
  const size_t wn = (size_t)4 * wstride, an = 4096;
  std::mt19937 rng(20261003);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (int r = 0; r < 2; ++r) {
    std::vector<uint16_t> hw(wn * 8), ha(an * 8);
    for (auto& v : hw) v = r ? f2h(nd(rng)) : f2h(0.0625f);
    for (auto& v : ha) v = r ? f2h(nd(rng)) : f2h(0.0625f);
    hipMalloc(&b[r].w, wn * 16);
    hipMalloc(&b[r].a, an * 16);
    hipMalloc(&b[r].out, sizeof(float) * 512 * 256);
    hipMemcpy(b[r].w, hw.data(), wn * 16, hipMemcpyHostToDevice);
    hipMemcpy(b[r].a, ha.data(), an * 16, hipMemcpyHostToDevice);
    b[r].wstride = wstride;
  }
  {   // idle
    Sampler smp(sens);
    std::this_thread::sleep_for(std::chrono::milliseconds(1000));
    Stats st = smp.finish();
    printf("idle: %.0f W, %.0f MHz\n\n", st.w_mean, st.mhz_mean);
  }
  printf("| operand feed (per 16-deep step and wave: 12 MFMA 32x32x16 f16) | operand data | TFLOP/s | of 2.5 PF | W mean | W max "
         "| sclk MHz mean | sclk MHz min | MFMA-busy MHz per SIMD | samples |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|\n");
  const char* dn[2] = {"constant 2^-4", "random fp16 (normal)"};
  const int blocks = prop.multiProcessorCount * 2;   // two 4-wave workgroups per CU = 2 waves per SIMD
  for (int r = 0; r < 2; ++r) {
    run<4, false, false>("registers only (ring of 4 operand sets)", dn[r], b[r], sens, seconds, blocks);
    run<4, true, false>("+ A fragments from LDS (4 x ds_read_b128)", dn[r], b[r], sens, seconds, blocks);
    run<4, false, true>("+ W fragments from L2 (4 x global_load_dwordx4)", dn[r], b[r], sens, seconds, blocks);
    run<4, true, true>("+ both: A from LDS, W from L2 (the product loop)", dn[r], b[r], sens, seconds, blocks);
  }
  // one workgroup per CU (one wave per SIMD), the product loop, random data: what half the occupancy costs
  run<4, true, true>("+ both, ONE wave per SIMD", dn[1], b[1], sens, seconds, prop.multiProcessorCount);
  // the single-plane bf16 form (configs[4]): random bf16 operands (the same bit patterns read as bf16 are not normal
  // numbers: a buffer of its own)
  {
    Bufs bb = b[1];
    std::vector<uint16_t> hw(wn * 8), ha(an * 8);
    auto f2bf = [](float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
    for (auto& v : hw) v = f2bf(nd(rng));
    for (auto& v : ha) v = f2bf(nd(rng));
    hipMemcpy(bb.w, hw.data(), wn * 16, hipMemcpyHostToDevice);
    hipMemcpy(bb.a, ha.data(), an * 16, hipMemcpyHostToDevice);
    printf("\n| bf16 single-plane loop (per step and wave: 8 MFMA 32x32x16 bf16, 4 row blocks x 64 channels) | operand data | TFLOP/s | of 2.5 PF "
           "| W mean | W max | sclk MHz mean | sclk MHz min | MFMA-busy MHz per SIMD | samples |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|\n");
    run<2, false, false, true>("registers only (ring of 2 operand sets)", "random bf16 (normal)", bb, sens, seconds, blocks);
    run<2, true, false, true>("+ A fragments from LDS (4 x ds_read_b128)", "random bf16 (normal)", bb, sens, seconds, blocks);
    run<2, false, true, true>("+ W fragments from L2 (2 x global_load_dwordx4)", "random bf16 (normal)", bb, sens, seconds, blocks);
    run<2, true, true, true>("+ both: the bf16 chains' loop", "random bf16 (normal)", bb, sens, seconds, blocks);
  }
  return 0;
}
