/*
 * drt_main.c -- entry point of the POSIX + HIP host (replaces src/win32_main.c:123-156):
 * read config.cfg from the working directory (or argv[1]), render, write the .spd files.
 */
#include "drt_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    const char *config_path = argc > 1 ? argv[1] : "config.cfg";
    FILE *f = fopen(config_path, "rb");
    if (!f)
    {
        fprintf(stderr, "cannot open %s\n", config_path);
        return 1;
    }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buffer = (char *)calloc((size_t)size + 1, 1);
    size_t got = fread(buffer, 1, (size_t)size, f);
    fclose(f);

    config_arguments args;
    memset(&args, 0, sizeof(args));
    parse_config(buffer, (u32)got, &args);
    free(buffer);

    printf("CONFIG ARGS:\n");
    print_config_arguments(&args);
    printf("\nStarting render...\n");
    render_image(&args);
    printf("Render complete.\n");
    return 0;
}
