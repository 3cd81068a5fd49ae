/*
 * png_codec.c -- read_png / write_png of the host shell, same contract as the
 * reference's hpc/read_img.c:9-65 and hpc/write_img.c:5-53 but without libpng
 * (the build image only has zlib headers): a small PNG codec over zlib.
 *
 * read:  8-bit gray, RGB and RGBA, non-interlaced. Colour is converted to gray
 *        with libpng's default rgb_to_gray weights (what png_set_rgb_to_gray(png, 1, -1, -1)
 *        at hpc/read_img.c:47-50 selects): (6968 R + 23434 G + 2366 B) >> 15, and, when
 *        the file carries an sRGB or gAMA chunk, through libpng's 8-bit gamma tables
 *        (linearise, weight with +16384 rounding, re-encode) exactly as libpng 1.6 does.
 *        Both forms are pinned against real libpng output in tests/golden/ (files ending in _gray_libpng.png).
 *        Anything else (gray+alpha, 16-bit, palette, interlaced) is rejected with -1
 *        instead of being silently misread (survey quirk Q13).
 * write: 8-bit gray, non-interlaced (hpc/write_img.c:38-45).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "glf.h"

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put_be32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

static const uint8_t PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

static int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return (pb <= pc) ? b : c;
}

/* want_rgb: rows of 3 * width bytes (R, G, B interleaved; a gray file replicates its value) instead of the gray conversion */
static int read_png_impl(const char *filename, uint8_t ***row_pointers, int *width, int *height, int want_rgb)
{
    if (!filename || !row_pointers || !width || !height) return -1;
    *row_pointers = NULL;
    FILE *f = fopen(filename, "rb");
    if (!f) {
        fprintf(stderr, "Could not open file %s\n", filename); /* hpc/read_img.c:16 */
        return -1;
    }
    int rc = -1;
    uint8_t *file = NULL, *zdata = NULL, *raw = NULL;
    uint8_t **rows = NULL;
    long fsize = 0;
    uint32_t w = 0, h = 0;
    int bit_depth = 0, color_type = -1, interlace = 0, seen_ihdr = 0, seen_iend = 0, channels = 0;
    size_t zlen = 0, zcap = 0, pos = 8, stride = 0, rawlen = 0;
    uLongf outlen = 0;
    long file_gamma = 0; /* libpng fixed point (x 100000); 0 = unknown */
    int have_srgb = 0;
    if (fseek(f, 0, SEEK_END) != 0 || (fsize = ftell(f)) < 8 + 25 || fseek(f, 0, SEEK_SET) != 0) goto done;
    file = (uint8_t *)malloc((size_t)fsize);
    if (!file || fread(file, 1, (size_t)fsize, f) != (size_t)fsize) goto done;
    if (memcmp(file, PNG_SIG, 8) != 0) goto done;

    while (pos + 12 <= (size_t)fsize && !seen_iend) {
        const uint32_t len = be32(file + pos);
        const uint8_t *type = file + pos + 4;
        if (len > 0x7fffffffu || pos + 12 + (size_t)len > (size_t)fsize) goto done;
        const uint8_t *data = file + pos + 8;
        const uint32_t crc = be32(data + len);
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != crc) goto done;
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) goto done;
            w = be32(data); h = be32(data + 4);
            bit_depth = data[8]; color_type = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) goto done;
            seen_ihdr = 1;
        } else if (memcmp(type, "IDAT", 4) == 0) {
            if (!seen_ihdr) goto done;
            if (zlen + len > zcap) {
                zcap = (zlen + len) * 2 + 4096;
                uint8_t *nz = (uint8_t *)realloc(zdata, zcap);
                if (!nz) goto done;
                zdata = nz;
            }
            memcpy(zdata + zlen, data, len);
            zlen += len;
        } else if (memcmp(type, "sRGB", 4) == 0) {
            have_srgb = 1;
            file_gamma = 45455; /* PNG_GAMMA_sRGB_INVERSE */
        } else if (memcmp(type, "gAMA", 4) == 0 && len == 4) {
            if (!have_srgb) file_gamma = (long)be32(data);
        } else if (memcmp(type, "IEND", 4) == 0) {
            seen_iend = 1;
        } else if (!(type[0] & 0x20)) {
            if (memcmp(type, "PLTE", 4) != 0) goto done; /* unknown critical chunk */
        }
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || !seen_iend || w == 0 || h == 0 || w > 65535u || h > 65535u) goto done;
    if (bit_depth != 8 || interlace != 0) goto done;
    switch (color_type) {
    case 0: channels = 1; break; /* gray */
    case 2: channels = 3; break; /* RGB  -> gray */
    case 6: channels = 4; break; /* RGBA -> gray, alpha dropped */
    default: goto done;          /* gray+alpha, palette: the reference misreads them (Q13) */
    }
    stride = (size_t)w * channels;
    rawlen = (stride + 1) * (size_t)h;
    raw = (uint8_t *)malloc(rawlen);
    if (!raw) goto done;
    outlen = (uLongf)rawlen;
    if (uncompress(raw, &outlen, zdata, (uLong)zlen) != Z_OK || outlen != rawlen) goto done;

    /* undo the scanline filters in place */
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *cur = raw + (stride + 1) * y + 1;
        const uint8_t *prev = y ? cur - (stride + 1) : NULL;
        const int ft = cur[-1];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)channels ? cur[x - channels] : 0;
            const int b = prev ? prev[x] : 0;
            const int c = (prev && x >= (size_t)channels) ? prev[x - channels] : 0;
            int v = cur[x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: goto done;
            }
            cur[x] = (uint8_t)v;
        }
    }

    /* libpng's 8-bit gamma tables for rgb_to_gray (png_build_8bit_table /
     * png_gamma_8bit_correct): only when the gamma is known and significant */
    uint8_t to1[256], from1[256];
    const int use_gamma = channels > 1 && file_gamma > 0 && labs(file_gamma - 100000) >= 5000;
    if (use_gamma) {
        const double recip = floor(1e10 / (double)file_gamma + .5) * 1e-5, g = (double)file_gamma * 1e-5;
        for (int i = 0; i < 256; ++i) {
            to1[i] = (uint8_t)floor(255. * pow(i / 255., recip) + .5);
            from1[i] = (uint8_t)floor(255. * pow(i / 255., g) + .5);
        }
    }

    /* same ownership as hpc/read_img.c:55-59: an array of malloc'd rows */
    rows = (uint8_t **)calloc(h, sizeof(uint8_t *));
    if (!rows) goto done;
    for (uint32_t y = 0; y < h; ++y) {
        rows[y] = (uint8_t *)malloc(want_rgb ? 3 * (size_t)w : w);
        if (!rows[y]) goto done;
        const uint8_t *src = raw + (stride + 1) * y + 1;
        if (want_rgb) {
            for (uint32_t x = 0; x < w; ++x)
                for (int ch = 0; ch < 3; ++ch) rows[y][3 * x + ch] = channels == 1 ? src[x] : src[x * channels + ch];
        } else if (channels == 1) memcpy(rows[y], src, w);
        else
            for (uint32_t x = 0; x < w; ++x) {
                const uint32_t r = src[x * channels], g = src[x * channels + 1], b = src[x * channels + 2];
                /* libpng png_do_rgb_to_gray, 8-bit, no gamma: equal channels pass through,
                 * otherwise the truncating weighted sum with the default coefficients */
                if (r == g && r == b) rows[y][x] = (uint8_t)r;
                else if (use_gamma)
                    rows[y][x] = from1[(6968u * to1[r] + 23434u * to1[g] + 2366u * to1[b] + 16384u) >> 15];
                else rows[y][x] = (uint8_t)((6968u * r + 23434u * g + 2366u * b) >> 15);
            }
    }
    *row_pointers = rows;
    *width = (int)w;
    *height = (int)h;
    rows = NULL;
    rc = 0;
done:
    if (rows) {
        for (uint32_t y = 0; y < h; ++y) free(rows[y]);
        free(rows);
    }
    free(raw);
    free(zdata);
    free(file);
    fclose(f);
    return rc;
}

int glf_read_png(const char *filename, uint8_t ***row_pointers, int *width, int *height)
{
    return read_png_impl(filename, row_pointers, width, height, 0);
}

int glf_read_png_rgb(const char *filename, uint8_t ***row_pointers, int *width, int *height)
{
    return read_png_impl(filename, row_pointers, width, height, 1);
}

static int write_chunk(FILE *f, const char *type, const uint8_t *data, uint32_t len)
{
    uint8_t hdr[8], crcb[4];
    put_be32(hdr, len);
    memcpy(hdr + 4, type, 4);
    uLong crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)type, 4);
    if (len) crc = crc32(crc, data, len);
    put_be32(crcb, (uint32_t)crc);
    if (fwrite(hdr, 1, 8, f) != 8) return -1;
    if (len && fwrite(data, 1, len, f) != len) return -1;
    if (fwrite(crcb, 1, 4, f) != 4) return -1;
    return 0;
}

static int write_png_impl(const char *filename, uint8_t **img_bytes, unsigned width, unsigned height, int rgb);

int glf_write_png(const char *filename, uint8_t **img_bytes, unsigned width, unsigned height)
{
    return write_png_impl(filename, img_bytes, width, height, 0);
}

int glf_write_png_rgb(const char *filename, uint8_t **img_bytes, unsigned width, unsigned height)
{
    return write_png_impl(filename, img_bytes, width, height, 1);
}

static int write_png_impl(const char *filename, uint8_t **img_bytes, unsigned width, unsigned height, int rgb)
{
    if (!filename || !img_bytes || width == 0 || height == 0) return -1;
    FILE *f = fopen(filename, "wb");
    if (!f) {
        fprintf(stderr, "Could not open file %s\n", filename); /* hpc/write_img.c:10 */
        return -1;
    }
    int rc = -1;
    const size_t rowbytes = (size_t)width * (rgb ? 3 : 1);
    const size_t rawlen = (rowbytes + 1) * height;
    uint8_t *raw = (uint8_t *)malloc(rawlen), *z = NULL;
    uLongf zlen = compressBound((uLong)rawlen);
    uint8_t ihdr[13];
    if (!raw) goto done;
    for (unsigned y = 0; y < height; ++y) {
        raw[(rowbytes + 1) * y] = 0; /* filter type None */
        memcpy(raw + (rowbytes + 1) * y + 1, img_bytes[y], rowbytes);
    }
    z = (uint8_t *)malloc(zlen);
    if (!z || compress2(z, &zlen, raw, (uLong)rawlen, 6) != Z_OK) goto done;
    put_be32(ihdr, width);
    put_be32(ihdr + 4, height);
    ihdr[8] = 8;  /* bit depth,  hpc/write_img.c:39 */
    ihdr[9] = rgb ? 2 : 0; /* PNG_COLOR_TYPE_GRAY, :38 (2 = RGB for the colour path) */
    ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0; /* base compression/filter, no interlace, :42-44 */
    if (fwrite(PNG_SIG, 1, 8, f) != 8) goto done;
    if (write_chunk(f, "IHDR", ihdr, 13) != 0) goto done;
    if (write_chunk(f, "IDAT", z, (uint32_t)zlen) != 0) goto done;
    if (write_chunk(f, "IEND", NULL, 0) != 0) goto done;
    rc = 0;
done:
    free(z);
    free(raw);
    if (fclose(f) != 0) rc = -1;
    return rc;
}
