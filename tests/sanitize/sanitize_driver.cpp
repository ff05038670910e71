// sanitize_driver.cpp -- TEST-ONLY: runs the CPU oracle (fp64 and fp32) and the host emulation of the kernel source (4 and 16
// lanes per robot) over the scenarios tests/test_sanitizers.py wrote, in a build with -fsanitize=address,undefined.
// Any out-of-bounds access, use of an uninitialised index, signed overflow or misaligned access in the oracle or in the
// kernel math headers (etg_core.h / etg_core16.h / etg_layout.h as the emulation compiles them) aborts the run.
// Scenario file: int32 count, then per scenario: EtgConfig bytes, EtgRobotModel bytes, int32 steps, heightfield floats
// (hf_nx * hf_ny, only when terrain == 1).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/etgsim.h"

extern "C" {
void* etgo_create64(const EtgConfig*, const EtgRobotModel*);
void etgo_destroy64(void*);
void etgo_set_params64(void*, const double*, const double*, const double*, int, const uint8_t*);
void etgo_set_heightfield64(void*, const float*);
void etgo_reset64(void*, const uint8_t*, double*, int);
void etgo_step64(void*, const double*, const uint8_t*, double*, double*, uint8_t*, double*, int);
void* etgo_create32(const EtgConfig*, const EtgRobotModel*);
void etgo_destroy32(void*);
void etgo_set_params32(void*, const float*, const float*, const float*, int, const uint8_t*);
void etgo_set_heightfield32(void*, const float*);
void etgo_reset32(void*, const uint8_t*, float*, int);
void etgo_step32(void*, const float*, const uint8_t*, float*, float*, uint8_t*, float*, int);
void* emu_create(const EtgConfig*, const EtgRobotModel*);
void emu_destroy(void*);
void emu_set_lanes(void*, int);
void emu_set_params(void*, const float*, const float*, const float*, int, const uint8_t*);
void emu_set_heightfield(void*, const float*);
void emu_reset(void*, const uint8_t*, float*);
void emu_step(void*, const float*, const uint8_t*, float*, float*, uint8_t*, float*);
}

static std::vector<float> default_dyn(int n) {   // param2dynamic_dict(zeros(48)), train.py:112-126
  std::vector<float> row(ETG_DYN_DIM, 1.0f);
  row[0] = 40.0f; row[1] = 0.2f; row[2] = 1.5f;
  for (int j = 0; j < 12; j++) { row[21 + j] = 80.0f; row[33 + j] = (j % 3 == 0) ? 1.0f : 2.0f; }
  row[45] = 0.0f; row[46] = 0.0f; row[47] = -10.0f;
  std::vector<float> all;
  for (int i = 0; i < n; i++) all.insert(all.end(), row.begin(), row.end());
  return all;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t count = 0;
  if (std::fread(&count, 4, 1, f) != 1) return 2;
  unsigned lcg = 12345u;
  auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((lcg >> 8) & 0xFFFF) / 65535.0f - 0.5f; };
  for (int s = 0; s < count; s++) {
    EtgConfig cfg;
    EtgRobotModel model;
    int32_t steps = 0;
    if (std::fread(&cfg, sizeof(cfg), 1, f) != 1 || std::fread(&model, sizeof(model), 1, f) != 1 || std::fread(&steps, 4, 1, f) != 1) return 2;
    std::vector<float> hf;
    if (cfg.terrain == 1) {
      hf.resize((size_t)cfg.hf_nx * cfg.hf_ny);
      if (std::fread(hf.data(), 4, hf.size(), f) != hf.size()) return 2;
    }
    const int n = cfg.num_envs, adim = cfg.motor_mode == 2 ? ETG_HYBRID_DIM : 12;
    std::vector<float> dynf = default_dyn(n), w(60, 0.0f), b(3, 0.0f);
    for (int k = 0; k < 20; k++) { w[k] = 0.02f * rnd(); w[40 + k] = 0.02f * rnd(); }
    std::vector<double> dynd(dynf.begin(), dynf.end()), wd(w.begin(), w.end()), bd(b.begin(), b.end());
    std::vector<float> actf((size_t)n * adim), obsf((size_t)n * ETG_OBS_DIM), rewf(n), inff((size_t)n * ETG_INFO_DIM);
    std::vector<double> actd((size_t)n * adim), obsd((size_t)n * ETG_OBS_DIM), rewd(n), infd((size_t)n * ETG_INFO_DIM);
    std::vector<uint8_t> done(n);
    void* o64 = etgo_create64(&cfg, &model);
    void* o32 = etgo_create32(&cfg, &model);
    void* e4 = emu_create(&cfg, &model);
    void* e16 = emu_create(&cfg, &model);
    emu_set_lanes(e4, 4);
    emu_set_lanes(e16, 16);
    const bool lanes4_ok = true;                        // (body rows: both mappings carry them)
    const bool lanes16_ok = cfg.body_contacts != 3;     // three body rows per leg: the 4-lane mapping only
    etgo_set_params64(o64, dynd.data(), wd.data(), bd.data(), 0, nullptr);
    etgo_set_params32(o32, dynf.data(), w.data(), b.data(), 0, nullptr);
    emu_set_params(e4, dynf.data(), w.data(), b.data(), 0, nullptr);
    emu_set_params(e16, dynf.data(), w.data(), b.data(), 0, nullptr);
    if (cfg.terrain == 1) {
      etgo_set_heightfield64(o64, hf.data()); etgo_set_heightfield32(o32, hf.data());
      emu_set_heightfield(e4, hf.data()); emu_set_heightfield(e16, hf.data());
    }
    etgo_reset64(o64, nullptr, obsd.data(), 1);
    etgo_reset32(o32, nullptr, obsf.data(), 1);
    if (lanes4_ok) emu_reset(e4, nullptr, obsf.data());
    if (lanes16_ok) emu_reset(e16, nullptr, obsf.data());
    for (int k = 0; k < steps; k++) {
      for (size_t i = 0; i < actf.size(); i++) { actf[i] = (cfg.motor_mode == 1 ? 8.0f : 0.3f) * rnd(); actd[i] = actf[i]; }
      if (cfg.motor_mode == 2)
        for (int i = 0; i < n * 12; i++) { actf[5 * i + 1] = 60.0f; actf[5 * i + 3] = 1.0f; actd[5 * i + 1] = 60.0; actd[5 * i + 3] = 1.0; }
      etgo_step64(o64, actd.data(), nullptr, obsd.data(), rewd.data(), done.data(), infd.data(), 1);
      etgo_step32(o32, actf.data(), nullptr, obsf.data(), rewf.data(), done.data(), inff.data(), 1);
      if (lanes4_ok) emu_step(e4, actf.data(), nullptr, obsf.data(), rewf.data(), done.data(), inff.data());
      if (lanes16_ok) emu_step(e16, actf.data(), nullptr, obsf.data(), rewf.data(), done.data(), inff.data());
    }
    etgo_destroy64(o64); etgo_destroy32(o32); emu_destroy(e4); emu_destroy(e16);
    std::printf("scenario %d ok (%d robots, %d steps)\n", s, n, steps);
  }
  std::fclose(f);
  std::printf("SANITIZE OK\n");
  return 0;
}
