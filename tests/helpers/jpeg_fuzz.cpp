// Mutation fuzzer for csrc/jpeg_core.h (host build).  Build with -fsanitize=address,undefined: any read or write outside a
// buffer, on any byte soup, aborts.   usage: jpeg_fuzz <iterations> <seed.jpg>...
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "jpeg_host.cpp"

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)n);
  if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
  fclose(f);
  return v;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const long iters = atol(argv[1]);
  std::vector<std::vector<uint8_t>> seeds;
  for (int i = 2; i < argc; ++i) { auto v = slurp(argv[i]); if (!v.empty()) seeds.push_back(v); }
  if (seeds.empty()) return 2;
  long decoded = 0, rejected = 0;
  for (long it = 0; it < iters; ++it) {
    std::vector<uint8_t> f = seeds[rnd() % seeds.size()];
    const int kind = rnd() % 6;
    if (kind == 0) {                                   // flip a few bytes anywhere
      for (int k = 0, n = 1 + rnd() % 8; k < n; ++k) f[rnd() % f.size()] = (uint8_t)rnd();
    } else if (kind == 1) {                            // flip bytes in the header region (tables, sizes, sampling factors)
      const size_t hdr = f.size() < 700 ? f.size() : 700;
      for (int k = 0, n = 1 + rnd() % 6; k < n; ++k) f[rnd() % hdr] = (uint8_t)rnd();
    } else if (kind == 2) {                            // truncate
      f.resize(1 + rnd() % f.size());
    } else if (kind == 3) {                            // sprinkle markers / 0xFF bytes into the entropy data
      for (int k = 0, n = 1 + rnd() % 6; k < n; ++k) { const size_t p = f.size() / 2 + rnd() % (f.size() / 2); f[p] = 0xFF; if (p + 1 < f.size() && (rnd() & 1)) f[p + 1] = (uint8_t)(0xC0 + rnd() % 0x40); }
    } else if (kind == 4) {                            // duplicate a slice (repeated segments, shifted offsets)
      const size_t a = rnd() % f.size(), n = rnd() % 300;
      std::vector<uint8_t> g(f.begin(), f.begin() + a);
      g.insert(g.end(), f.begin() + a, f.begin() + (a + n < f.size() ? a + n : f.size()));
      g.insert(g.end(), f.begin() + a, f.end());
      f.swap(g);
    } else {                                           // random bytes behind a valid SOI
      for (size_t k = 2; k < f.size(); ++k) if ((rnd() & 7) == 0) f[k] = (uint8_t)rnd();
    }
    const long long r = jpeg_host_decode_checked(f.data(), (int64_t)f.size(), 1 << 22);
    if (r >= 0) ++decoded; else ++rejected;
  }
  printf("iterations %ld: decoded %ld, rejected %ld\n", iters, decoded, rejected);
  return 0;
}
