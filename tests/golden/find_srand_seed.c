/* Which srand(time(0)) coloured line_matching/data/edline_result.png?
 *
 * The reference's demo (line_matching/src/test_edline_detector.cpp:42-59) seeds rand() with the wall clock and gives
 * line i of the detector's output list the colour  r = rand() % 256, g = rand() % 256, b = rand() % 256,
 * cv::Scalar(r, g, b)  -- B, G, R, so the picture's (R, G, B) is (third, second, first).  glibc's rand() is a published
 * additive-feedback generator (TYPE_3: r[i] = r[i-3] + r[i-31], seeded by 16807 * x mod 2^31-1, 310 values discarded,
 * result >> 1), restated below.  Given the colours of the stripes recovered from the picture
 * (edline_result_segments.npz: colour_rgb), a seed fits if its first 12 colours are all among them; one seed in
 * [2015, 2026) does, and with it all 258 colours of the sequence are the 258 stripes' -- which gives every stripe its
 * POSITION in the reference's output list (make_edline_result_segments.py stores it as list_index).
 *
 *   python -c "import numpy as np; [print(int(r)<<16|int(g)<<8|int(b)) for r,g,b in np.load('edline_result_segments.npz')['colour_rgb']]" > cols.txt
 *   gcc -O2 -o find_srand_seed find_srand_seed.c && ./find_srand_seed cols.txt 1420000000 1780000000
 *   -> seed 1612579976        (Sat Feb  6 2021; about 20 s on 8 cores when the range is split)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static unsigned char *present;   /* bit set over 24-bit colours */

static int has(unsigned k) { return present[k >> 3] >> (k & 7) & 1; }

static void glibc_rand(uint32_t seed, int n, int *out) {
    int32_t r[344 + 64];
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        int64_t v = (16807LL * r[i - 1]) % 2147483647;
        if (v < 0) v += 2147483647;
        r[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 34; i < 344 + n; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    for (int i = 0; i < n; i++) out[i] = (int)(((uint32_t)r[344 + i]) >> 1);
}

int main(int argc, char **argv) {
    if (argc < 4) return fprintf(stderr, "usage: %s colours.txt first_seed last_seed\n", argv[0]), 2;
    present = calloc(1 << 21, 1);
    FILE *f = fopen(argv[1], "r");
    if (!f) return perror(argv[1]), 2;
    unsigned k;
    while (fscanf(f, "%u", &k) == 1) present[(k & 0xFFFFFF) >> 3] |= 1 << (k & 7);
    fclose(f);
    uint32_t lo = strtoul(argv[2], 0, 10), hi = strtoul(argv[3], 0, 10);
    int o[36];
    for (uint32_t s = lo; s < hi; s++) {
        glibc_rand(s, 36, o);
        int ok = 1;
        for (int i = 0; i < 12 && ok; i++) {
            unsigned a = o[3 * i] & 255, b = o[3 * i + 1] & 255, c = o[3 * i + 2] & 255;
            ok = has(c << 16 | b << 8 | a);
        }
        if (ok) printf("seed %u\n", s), fflush(stdout);
    }
    return 0;
}
