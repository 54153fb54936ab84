// Times builds of csrc/xdt_proj.hip against each other (tuning tool; no torch).
//   hipcc --offload-arch=gfx950 -O3 tools/xdt_probe.hip -o tools/xdt_probe -ldl
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC [-DXDT_...] si_mamba_amd/csrc/xdt_proj.hip -o tools/alt/xdt_<v>.so
//   ./tools/xdt_probe tools/alt/xdt_a.so tools/alt/xdt_b.so ...
// Every library runs simamba_conv_xdt_proj_fwd and simamba_xdt_proj_fwd at (B, 768, 1024, S = 56, R = 24) on the x half
// of a (B, 2 D, L) buffer; outputs are compared bit for bit with the first library's.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int (*conv_fn)(const void*, const float*, const float*, const float*, const float*, void*, void*, void*, int, int,
                       int, int, int, int, long long, void*);
typedef int (*plain_fn)(const void*, const float*, const float*, void*, void*, int, int, int, int, int, int, long long,
                        void*);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void fill(std::vector<float>& v, unsigned seed, float scale) {
  unsigned s = seed;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = scale * (static_cast<int>(s >> 8) % 20001 - 10000) * 1e-4f; }
}

int main(int argc, char** argv) {
  const int D = 768, L = 1024, S = 56, R = 24;
  const int Bs[2] = {64, 128};
  const int Bmax = 128;
  const size_t nxz = static_cast<size_t>(Bmax) * 2 * D * L, nx = static_cast<size_t>(Bmax) * D * L;
  std::vector<float> hxz(nxz), hcw(D * 4), hcb(D), hwx(S * D), hwdt(D * R);
  fill(hxz, 1, 1.f); fill(hcw, 2, .5f); fill(hcb, 3, .5f); fill(hwx, 4, .04f); fill(hwdt, 5, .2f);
  float *xz, *cw, *cb, *wx, *wdt, *xc, *xdbl, *delta;
  CK(hipMalloc(&xz, nxz * 4)); CK(hipMalloc(&cw, D * 16)); CK(hipMalloc(&cb, D * 4)); CK(hipMalloc(&wx, S * D * 4));
  CK(hipMalloc(&wdt, D * R * 4)); CK(hipMalloc(&xc, nx * 4)); CK(hipMalloc(&delta, nx * 4));
  CK(hipMalloc(&xdbl, static_cast<size_t>(Bmax) * L * S * 4));
  CK(hipMemcpy(xz, hxz.data(), nxz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(cw, hcw.data(), D * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(cb, hcb.data(), D * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wx, hwx.data(), S * D * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wdt, hwdt.data(), D * R * 4, hipMemcpyHostToDevice));
  const size_t ncmp = static_cast<size_t>(64) * D * L;
  std::vector<float> ref_delta[2], ref_xc, ref_xdbl[2], got(ncmp);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // clocks up before the first timed library
  for (int i = 0; i < 400; ++i) CK(hipMemsetAsync(delta, 0, nx * 4));
  CK(hipDeviceSynchronize());
  for (int a = 1; a < argc; ++a) {
    void* h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("%s: %s\n", argv[a], dlerror()); return 1; }
    conv_fn fc = reinterpret_cast<conv_fn>(dlsym(h, "simamba_conv_xdt_proj_fwd"));
    plain_fn fp = reinterpret_cast<plain_fn>(dlsym(h, "simamba_xdt_proj_fwd"));
    for (int bi = 0; bi < 2; ++bi) {
      const int B = Bs[bi];
      for (int mode = 0; mode < 2; ++mode) {
        auto run = [&]() {
          return mode == 0 ? fc(xz, cw, cb, wx, wdt, xc, xdbl, delta, B, D, L, S, R, 0, 2LL * D * L, nullptr)
                           : fp(xz, wx, wdt, xdbl, delta, B, D, L, S, R, 0, 2LL * D * L, nullptr);
        };
        CK(hipMemset(delta, 0xff, nx * 4));
        for (int i = 0; i < 3; ++i) { int rc = run(); if (rc) { printf("rc %d\n", rc); return 1; } }
        CK(hipEventRecord(e0));
        const int it = 20;
        for (int i = 0; i < it; ++i) run();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* tag = "";
        if (bi == 0) {
          size_t bad = 0;
          CK(hipMemcpy(got.data(), delta, ncmp * 4, hipMemcpyDeviceToHost));
          if (a == 1) ref_delta[mode] = got; else bad += memcmp(got.data(), ref_delta[mode].data(), ncmp * 4) != 0;
          if (mode == 0) {
            CK(hipMemcpy(got.data(), xc, ncmp * 4, hipMemcpyDeviceToHost));
            if (a == 1) ref_xc = got; else bad += 2 * (memcmp(got.data(), ref_xc.data(), ncmp * 4) != 0);
          }
          const size_t nd = static_cast<size_t>(64) * L * S;
          CK(hipMemcpy(got.data(), xdbl, nd * 4, hipMemcpyDeviceToHost));
          if (a == 1) ref_xdbl[mode].assign(got.begin(), got.begin() + nd); else bad += 4 * (memcmp(got.data(), ref_xdbl[mode].data(), nd * 4) != 0);
          static char tb[64]; snprintf(tb, sizeof tb, " OUTPUTS DIFFER (mask %zu)", bad);
          tag = a == 1 ? " (reference outputs)" : (bad ? tb : " bit-identical");
        }
        printf("%-28s B=%3d %-5s %8.1f us%s\n", argv[a], B, mode == 0 ? "conv" : "plain", ms * 1e3 / it, tag);
        fflush(stdout);
        typedef int (*dump_fn)(long long*);
        dump_fn dump = reinterpret_cast<dump_fn>(dlsym(h, "xdt_debug_dump"));
        if (dump && bi == 0) {
          static long long d[4][64][8];
          run(); CK(hipDeviceSynchronize());
          dump(&d[0][0][0]);
          for (int sl = 0; sl < 4; ++sl) {
            printf("  slot %d (block %d wave %d): per iteration: start->reads issued->mfma issued->staged->units->loads issued->pre-barrier->post-barrier (cycles since iteration start), total\n", sl, sl < 2 ? 0 : 300, sl & 1 ? 3 : 0);
            for (int g = 0; g < 52; ++g) {
              printf("   g=%2d", g);
              for (int k = 1; k < 8; ++k) printf(" %6lld", d[sl][g][k] - d[sl][g][0]);
              if (g + 1 < 64) printf("   | %6lld", d[sl][g + 1][0] - d[sl][g][0]);
              printf("\n");
            }
          }
        }
      }
    }
  }
  return 0;
}
