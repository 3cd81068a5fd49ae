// host_util.cpp -- host-side stages of the C-ABI: sampling grid, X0 random block,
// synthetic benchmark images. No device code.
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "../../include/glf.h"

namespace {

inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

struct Xoshiro256ss {
    uint64_t s[4];
    explicit Xoshiro256ss(uint64_t seed)
    {
        // splitmix64 expansion of the seed
        for (int i = 0; i < 4; ++i) {
            uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[i] = z ^ (z >> 31);
        }
    }
    uint64_t next()
    {
        const uint64_t result = rotl64(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl64(s[3], 45);
        return result;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); } // [0,1)
};

} // namespace

extern "C" {

// Spatially uniform grid sampling, same contract as the reference's
// Sampling()/UniformSampling() (hpc/sampling.c:6-33): the requested count is
// rewritten to the realised grid count; indices ascend in raster order; the
// grid pitch is floor(sqrt(floor(N / requested))) with unsigned integer
// division; the last image row and column are never sampled.
int glf_Sampling(int width, int height, unsigned *sample_size, unsigned **sample_indices)
{
    // width or height 1: `r < h - 1u` would wrap (the reference's loop bound `i < height - 1`, hpc/sampling.c:16-18, admits no
    // row either: no sample exists) -- rejected instead of counting forever
    if (!sample_size || !sample_indices || width < 2 || height < 2 || *sample_size == 0)
        return GLF_ERR_INVALID;
    const unsigned w = (unsigned)width, h = (unsigned)height;
    const unsigned pitch = (unsigned)std::sqrt((double)((w * h) / *sample_size));
    if (pitch == 0) return GLF_ERR_INVALID;
    const unsigned first = pitch / 2;
    unsigned nrows = 0, ncols = 0;
    for (unsigned r = first; r < h - 1u; r += pitch) ++nrows;
    for (unsigned c = first; c < w - 1u; c += pitch) ++ncols;
    const size_t count = (size_t)nrows * ncols;
    unsigned *idx = (unsigned *)std::malloc(sizeof(unsigned) * (count ? count : 1));
    if (!idx) return GLF_ERR_NOMEM;
    size_t k = 0;
    for (unsigned r = first; r < h - 1u; r += pitch)
        for (unsigned c = first; c < w - 1u; c += pitch) idx[k++] = w * r + c;
    *sample_size = (unsigned)count;
    *sample_indices = idx;
    return GLF_OK;
}

int glf_shard_rows(int height, int rank, int size, int *row0, int *row1)
{
    if (height < 0 || size < 1 || rank < 0 || rank >= size || !row0 || !row1) return GLF_ERR_INVALID;
    *row0 = (int)((long long)rank * height / size);
    *row1 = (int)((long long)(rank + 1) * height / size);
    return GLF_OK;
}

// X0 for the inverse subspace iteration: m vectors of length p, vector after
// vector, U[0,1) (the reference fills with PETSc's rand48 seeded by the MPI
// rank, hpc/inverse_power_it.c:27-34; that stream is third-party, so we fix our
// own and make it independent of the GPU count).
int glf_random_vectors(double *X0, unsigned p, unsigned m, uint64_t seed)
{
    if (!X0) return GLF_ERR_INVALID;
    Xoshiro256ss rng(seed);
    const size_t n = (size_t)p * m;
    for (size_t k = 0; k < n; ++k) X0[k] = rng.uniform();
    return GLF_OK;
}

// Synthetic noisy benchmark image: smooth low-frequency shading +
// 64-px two-level checker + one diagonal edge, scaled into [40, 215], plus
// i.i.d. Gaussian noise sigma = 20 (Box-Muller), rounded and clipped to uint8.
int glf_synth_image(uint8_t *out, int width, int height, uint64_t seed)
{
    if (!out || width <= 0 || height <= 0) return GLF_ERR_INVALID;
    Xoshiro256ss rng(0x5EED0000ull + seed);
    const double two_pi = 6.283185307179586;
    for (int r = 0; r < height; ++r) {
        for (int c = 0; c < width; ++c) {
            const double u = (double)c / 512.0, v = (double)r / 512.0;
            double base = 0.5 * (std::sin(two_pi * 0.9 * u) + std::sin(two_pi * 0.6 * v + 1.0) +
                                 std::sin(two_pi * 0.4 * (u + v))) / 3.0;      // [-0.5, 0.5]
            base += (((r >> 6) + (c >> 6)) & 1) ? 0.22 : -0.22;                   // checker
            base += ((c - r) > (width - height) / 2 + 37) ? 0.15 : -0.15;         // diagonal edge
            double g = 127.5 + base * (175.0 / 1.74);                             // ~[40, 215]
            const double u1 = 1.0 - rng.uniform(), u2 = rng.uniform();
            g += 20.0 * std::sqrt(-2.0 * std::log(u1)) * std::cos(two_pi * u2);
            g = std::nearbyint(g);
            out[(size_t)r * width + c] = (uint8_t)(g < 0.0 ? 0.0 : (g > 255.0 ? 255.0 : g));
        }
    }
    return GLF_OK;
}

} // extern "C"
