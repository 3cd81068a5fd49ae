// host_util.cpp -- host-side stages of the C-ABI: sampling grid, X0 random block,
// synthetic benchmark images. No device code.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

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

// The PoC's random sampler (python/sampling/random.py:8-16): draw until `*sample_size` distinct pixels are there, sort.
int glf_RandomSampling(int width, int height, unsigned *sample_size, unsigned **sample_indices, uint64_t seed)
{
    if (!sample_size || !sample_indices || width <= 0 || height <= 0) return GLF_ERR_INVALID;
    const uint64_t N = (uint64_t)width * (uint64_t)height;
    const unsigned want = *sample_size;
    *sample_indices = nullptr;
    if (want == 0 || want > N) return GLF_ERR_INVALID;
    std::vector<uint8_t> taken(N, 0);
    Xoshiro256ss rng(0xA11CE5EEDull ^ seed);
    unsigned have = 0;
    while (have < want) {
        const uint64_t px = rng.next() % N;
        if (!taken[px]) {
            taken[px] = 1;
            ++have;
        }
    }
    unsigned *idx = static_cast<unsigned *>(std::malloc(sizeof(unsigned) * want));
    if (!idx) return GLF_ERR_NOMEM;
    unsigned k = 0;
    for (uint64_t px = 0; px < N; ++px)
        if (taken[px]) idx[k++] = (unsigned)px;
    *sample_indices = idx;
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

// ---- low-rank factor of the photometric table -----------------------------------------------------------------
// P[v][w] = exp2(-s_val (v - w)^2) over the 256 grey levels (the photometric factor of hpc/affinity.c:59-113 with
// s_val = log2(e) / h_val^2) is symmetric positive semi-definite with a rapidly decaying spectrum: at the reference's
// h_val = 30 its rank-32 eigen-expansion P ~= F F^T reproduces every entry to 1.5e-11. The factor is what the "rank"
// form of the grid-factored contractions carries instead of the 256 grey levels (nystroem_rank.inc).
// Computed in f64: pivoted Cholesky P ~= L L^T down to a residual diagonal of 1e-15 (r <= max_chol columns), then the
// eigen-decomposition of the small r x r matrix L^T L = Q diag(lambda) Q^T by cyclic Jacobi; F = L Q has the columns
// sqrt(lambda_k) u_k of the eigen-expansion, strongest first. Rows of F have sum_k F[v][k]^2 <= P[v][v] = 1.
namespace glf {

// F: [256][rank_out] row-major. Returns false when more than max_chol Cholesky columns are needed (sharp kernels).
bool photometric_factor(double s_val, int max_chol, std::vector<double> &F, int &rank_out)
{
    constexpr int n = 256;
    std::vector<double> L((size_t)n * max_chol, 0.0), d(n, 1.0);
    auto P = [&](int v, int w) { return std::exp2(-s_val * (double)(v - w) * (double)(v - w)); };
    int r = 0;
    for (; r < max_chol; ++r) {
        int piv = 0;
        for (int v = 1; v < n; ++v)
            if (d[v] > d[piv]) piv = v;
        if (d[piv] <= 1e-15) break;
        const double inv = 1.0 / std::sqrt(d[piv]);
        for (int v = 0; v < n; ++v) {
            double x = P(v, piv);
            for (int k = 0; k < r; ++k) x -= L[(size_t)v * max_chol + k] * L[(size_t)piv * max_chol + k];
            x *= inv;
            L[(size_t)v * max_chol + r] = x;
            d[v] -= x * x;
        }
        d[piv] = 0.0;
    }
    if (r == max_chol) {
        double worst = 0.0;
        for (int v = 0; v < n; ++v) worst = std::max(worst, d[v]);
        if (worst > 1e-15) return false;
    }
    // G = L^T L (r x r), Jacobi eigen-decomposition G = Q diag(lam) Q^T
    std::vector<double> G((size_t)r * r, 0.0), Q((size_t)r * r, 0.0);
    for (int i = 0; i < r; ++i)
        for (int j = i; j < r; ++j) {
            double x = 0.0;
            for (int v = 0; v < n; ++v) x += L[(size_t)v * max_chol + i] * L[(size_t)v * max_chol + j];
            G[(size_t)i * r + j] = G[(size_t)j * r + i] = x;
        }
    for (int i = 0; i < r; ++i) Q[(size_t)i * r + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < r; ++i) {
            diag += G[(size_t)i * r + i] * G[(size_t)i * r + i];
            for (int j = i + 1; j < r; ++j) off += G[(size_t)i * r + j] * G[(size_t)i * r + j];
        }
        if (off <= 1e-32 * diag) break;
        for (int pI = 0; pI < r - 1; ++pI)
            for (int q = pI + 1; q < r; ++q) {
                const double apq = G[(size_t)pI * r + q];
                if (apq == 0.0) continue;
                const double app = G[(size_t)pI * r + pI], aqq = G[(size_t)q * r + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < r; ++k) { // columns p, q
                    const double gkp = G[(size_t)k * r + pI], gkq = G[(size_t)k * r + q];
                    G[(size_t)k * r + pI] = c * gkp - s * gkq;
                    G[(size_t)k * r + q] = s * gkp + c * gkq;
                }
                for (int k = 0; k < r; ++k) { // rows p, q
                    const double gpk = G[(size_t)pI * r + k], gqk = G[(size_t)q * r + k];
                    G[(size_t)pI * r + k] = c * gpk - s * gqk;
                    G[(size_t)q * r + k] = s * gpk + c * gqk;
                }
                for (int k = 0; k < r; ++k) {
                    const double qkp = Q[(size_t)k * r + pI], qkq = Q[(size_t)k * r + q];
                    Q[(size_t)k * r + pI] = c * qkp - s * qkq;
                    Q[(size_t)k * r + q] = s * qkp + c * qkq;
                }
            }
    }
    std::vector<int> order(r);
    for (int i = 0; i < r; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return G[(size_t)a * r + a] > G[(size_t)b * r + b]; });
    F.assign((size_t)n * r, 0.0);
    for (int k = 0; k < r; ++k) {
        const int col = order[k];
        // sign convention: the entry of largest magnitude of every column is positive (the product F F^T does not care)
        double big = 0.0;
        for (int v = 0; v < n; ++v) {
            double x = 0.0;
            for (int j = 0; j < r; ++j) x += L[(size_t)v * max_chol + j] * Q[(size_t)j * r + col];
            F[(size_t)v * r + k] = x;
            if (std::fabs(x) > std::fabs(big)) big = x;
        }
        if (big < 0.0)
            for (int v = 0; v < n; ++v) F[(size_t)v * r + k] = -F[(size_t)v * r + k];
    }
    rank_out = r;
    return true;
}

// max over (v, w) of |sum_{k < R} F[v][k] F[w][k] - P[v][w]|
double photometric_factor_error(double s_val, const std::vector<double> &F, int rank, int R)
{
    constexpr int n = 256;
    double worst = 0.0;
    for (int v = 0; v < n; ++v)
        for (int w = v; w < n; ++w) {
            double x = 0.0;
            for (int k = 0; k < R && k < rank; ++k) x += F[(size_t)v * rank + k] * F[(size_t)w * rank + k];
            worst = std::max(worst, std::fabs(x - std::exp2(-s_val * (double)(v - w) * (double)(v - w))));
        }
    return worst;
}

} // namespace glf
