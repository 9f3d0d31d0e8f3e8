/*
 * drt_bmp.c -- .spd -> linear RGB -> BMP post-process, the tail of the reference's main():
 * spd_file_to_rgb_f64_pixels (src/daily_ray_trace.c:1-28), spectrum_to_xyz / spectrum_to_rgb_f64
 * (src/spectrum.c:49-82), rgb_f64_to_rgb_u8 and the BMP writer (src/win32_platform.c:11-41, :136-161).
 * Same arithmetic and byte layout: divide by the filter sum when the file carries one, XYZ from the
 * colour-matching tables, the fixed XYZ -> linear RGB matrix, clamp to [0,1], truncate to 8 bits (no gamma),
 * 32-bit BGRA, bottom-up, 3780 px/m. The alpha byte, which the reference leaves uninitialised, is 255.
 */
#include "drt_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#pragma pack(push, 1)
typedef struct { uint16_t bfType; uint32_t bfSize; uint16_t bfReserved1, bfReserved2; uint32_t bfOffBits; } bmp_file_header; /* 14 bytes */
typedef struct
{
    uint32_t biSize; int32_t biWidth, biHeight; uint16_t biPlanes, biBitCount; uint32_t biCompression, biSizeImage;
    int32_t biXPelsPerMeter, biYPelsPerMeter; uint32_t biClrUsed, biClrImportant;
} bmp_info_header; /* 40 bytes */
#pragma pack(pop)

static f64 clamp01(f64 f) /* clamp(), src/utils.c:6-11 */
{
    f = (f < 0.0) ? 0.0 : f;
    f = (f > 1.0) ? 1.0 : f;
    return f;
}

/* spectrum_to_xyz + spectrum_to_rgb_f64 with explicit tables: cmf = [4][S] rows rw, x, y, z */
void drt_host_spectrum_to_rgb(const f64 *cmf, u32 S, f64 interval, const f64 *spd, f64 rgb[3])
{
    const f64 *rw = cmf, *cx = cmf + S, *cy = cmf + 2 * (size_t)S, *cz = cmf + 3 * (size_t)S;
    f64 n = 0.0, X = 0.0, Y = 0.0, Z = 0.0;
    for (u32 i = 0; i < S; i += 1) n += (cy[i] * rw[i]);
    n *= interval;
    for (u32 i = 0; i < S; i += 1)
    {
        X += (cx[i] * spd[i] * rw[i]);
        Y += (cy[i] * spd[i] * rw[i]);
        Z += (cz[i] * spd[i] * rw[i]);
    }
    X *= (interval / n);
    Y *= (interval / n);
    Z *= (interval / n);
    rgb[0] = (2.3706743 * X) - (0.9000405 * Y) - (0.4706338 * Z);
    rgb[1] = (-0.5138850 * X) + (1.4253036 * Y) + (0.0885814 * Z);
    rgb[2] = (0.0052982 * X) - (0.0146949 * Y) + (1.0093968 * Z);
}

/* headers + 4 bytes per pixel (B, G, R, A; row 0 first), write_pixels_to_bmp (src/win32_platform.c:11-41) */
int drt_host_write_bmp_bgra(const char *path, u32 width, u32 height, const u8 *bgra)
{
    size_t n = (size_t)width * height;
    u8 head[sizeof(bmp_file_header) + sizeof(bmp_info_header)];
    memset(head, 0, sizeof(head));
    bmp_file_header *fh = (bmp_file_header *)head;
    bmp_info_header *ih = (bmp_info_header *)(head + sizeof(bmp_file_header));
    fh->bfType = 0x4d42;
    fh->bfSize = (uint32_t)(sizeof(head) + n * 4);
    fh->bfOffBits = sizeof(head);
    ih->biSize = sizeof(bmp_info_header);
    ih->biWidth = (int32_t)width;
    ih->biHeight = (int32_t)height; /* positive: bottom-up, and pixel row 0 is the bottom of the film */
    ih->biPlanes = 1;               /* the reference leaves 0 here; 1 is what the format requires */
    ih->biBitCount = 32;
    ih->biXPelsPerMeter = 3780;
    ih->biYPelsPerMeter = 3780;
    FILE *f = fopen(path, "wb");
    if (!f) return -2;
    int ok = fwrite(head, 1, sizeof(head), f) == sizeof(head) && fwrite(bgra, 1, n * 4, f) == n * 4;
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -3;
}

int drt_host_write_bmp(const char *path, u32 width, u32 height, const f64 *rgb /* [h*w][3], row 0 first */)
{
    size_t n = (size_t)width * height;
    u8 *px = (u8 *)malloc(n * 4 + 4);
    if (!px) return -1;
    for (size_t i = 0; i < n; i += 1)
    {
        px[4 * i + 2] = (u8)(clamp01(rgb[3 * i + 0]) * 255.0);
        px[4 * i + 1] = (u8)(clamp01(rgb[3 * i + 1]) * 255.0);
        px[4 * i + 0] = (u8)(clamp01(rgb[3 * i + 2]) * 255.0);
        px[4 * i + 3] = 255;
    }
    int rc = drt_host_write_bmp_bgra(path, width, height, px);
    free(px);
    return rc;
}

/* spd_file_to_bmp, src/win32_main.c:115-121 */
int drt_host_spd_file_to_bmp(const char *spd_path, const char *bmp_path, const f64 *cmf /* [4][S] */)
{
    spd_file_header header;
    f64 *pixels = NULL;
    int rc = drt_host_read_spd(spd_path, &header, &pixels);
    if (rc) return rc;
    u32 S = header.number_of_wavelengths;
    size_t per_pixel = (size_t)S + (header.has_filter_values ? 1 : 0);
    size_t n = (size_t)header.width_in_pixels * header.height_in_pixels;
    f64 *rgb = (f64 *)malloc(n * 3 * sizeof(f64));
    f64 *tmp = (f64 *)malloc((size_t)S * sizeof(f64));
    for (size_t i = 0; i < n; i += 1)
    {
        const f64 *p = pixels + i * per_pixel;
        const f64 *spd = p;
        if (header.has_filter_values)
        {
            f64 filter = p[S];
            for (u32 k = 0; k < S; k += 1) tmp[k] = p[k] / filter;
            spd = tmp;
        }
        drt_host_spectrum_to_rgb(cmf, S, header.wavelength_interval, spd, rgb + 3 * i);
    }
    rc = drt_host_write_bmp(bmp_path, header.width_in_pixels, header.height_in_pixels, rgb);
    free(tmp);
    free(rgb);
    free(pixels);
    return rc;
}
