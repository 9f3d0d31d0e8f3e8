/*
 * drt_spd_io.c -- the .spd file format (src/daily_ray_trace.h:59-68): 40-byte header, then row-major
 * pixels of S (+1 filter sum) doubles, y = 0 first.
 */
#include "drt_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* 40-byte header + row-major pixels, y = 0 first (src/daily_ray_trace.c:667-680, :758-770) */
int drt_host_write_spd(const char *path, u32 width, u32 height, u32 num_wl, u32 has_filter, f64 min_wl, f64 interval,
                       const f64 *pixels)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    spd_file_header header;
    memset(&header, 0, sizeof(header));
    header.id = 0xedfeefbe;
    header.width_in_pixels = width;
    header.height_in_pixels = height;
    header.number_of_wavelengths = num_wl;
    header.has_filter_values = has_filter;
    header.min_wavelength = min_wl;
    header.wavelength_interval = interval;
    size_t per_pixel = (size_t)num_wl + (has_filter ? 1 : 0);
    size_t count = (size_t)width * height * per_pixel;
    int ok = fwrite(&header, sizeof(header), 1, f) == 1 && fwrite(pixels, sizeof(f64), count, f) == count;
    fclose(f);
    return ok ? 0 : -1;
}

int drt_host_read_spd(const char *path, spd_file_header *header, f64 **pixels)
{
    *pixels = NULL;
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    if (fread(header, sizeof(*header), 1, f) != 1 || header->id != 0xedfeefbe) { fclose(f); return -2; }
    /* the header is not trusted: the pixel count it announces must be what the file holds */
    size_t per_pixel = (size_t)header->number_of_wavelengths + (header->has_filter_values ? 1 : 0);
    unsigned long long count = (unsigned long long)header->width_in_pixels * header->height_in_pixels * per_pixel;
    long at = ftell(f);
    fseek(f, 0, SEEK_END);
    long end = ftell(f);
    fseek(f, at, SEEK_SET);
    if (at < 0 || end < at || count > (unsigned long long)(end - at) / sizeof(f64)) { fclose(f); return -3; }
    *pixels = (f64 *)malloc((size_t)(count ? count : 1) * sizeof(f64));
    if (!*pixels) { fclose(f); return -4; }
    if (fread(*pixels, sizeof(f64), (size_t)count, f) != (size_t)count) { fclose(f); free(*pixels); *pixels = NULL; return -3; }
    fclose(f);
    return 0;
}
