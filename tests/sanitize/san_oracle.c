/* san_oracle.c -- ASan / UBSan driver for the CPU oracle (oracle/sparkzstd_oracle.c, the checker): the corpus
 * intact (must decode, into an exact-size destination), mutated and truncated (status or output, never a fault).
 * usage: san_oracle <n_mutations_per_frame> file.zst... */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/sparkzstd_oracle.h"

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd(void)
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int n_mut = atoi(argv[1]);
    unsigned long long n_ok = 0, n_bad = 0;
    for (int i = 2; i < argc; i++) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) return 2;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *src = (uint8_t *)malloc((size_t)n);
        if (fread(src, 1, (size_t)n, f) != (size_t)n) return 2;
        fclose(f);
        /* intact: learn the size with a roomy buffer, then decode into exactly that many bytes, with a trace */
        size_t cap = 8u << 20, out_len = 0, used = 0;
        uint8_t *dst = (uint8_t *)malloc(cap);
        int rc = orc_decode_frame(src, (size_t)n, dst, cap, &out_len, &used, NULL);
        if (rc != 0) { fprintf(stderr, "%s: intact frame failed: %s\n", argv[i], orc_strerror(rc)); return 1; }
        free(dst);
        dst = (uint8_t *)malloc(out_len ? out_len : 1);
        orc_trace tr;
        memset(&tr, 0, sizeof tr);
        size_t out2 = 0;
        rc = orc_decode_frame(src, (size_t)n, dst, out_len, &out2, &used, &tr);
        if (rc != 0 || out2 != out_len) { fprintf(stderr, "%s: exact-size decode failed\n", argv[i]); return 1; }
        orc_trace_free(&tr);
        /* a destination that is too small is an error, not an overflow */
        if (out_len > 1) {
            uint8_t *small = (uint8_t *)malloc(out_len - 1);
            rc = orc_decode_frame(src, (size_t)n, small, out_len - 1, &out2, &used, NULL);
            if (rc == 0) { fprintf(stderr, "%s: decoded into a short buffer\n", argv[i]); return 1; }
            free(small);
        }
        for (int m = 0; m < n_mut; m++) {
            size_t len = (size_t)n;
            uint8_t *b = (uint8_t *)malloc(len ? len : 1);
            memcpy(b, src, len);
            if (rnd() % 5 == 0) {
                len = (size_t)(rnd() % (uint64_t)(n + 1));
                uint8_t *t = (uint8_t *)malloc(len ? len : 1);  /* exact-size copy of the prefix */
                memcpy(t, b, len);
                free(b);
                b = t;
            } else {
                const size_t lo = (rnd() % 10 < 3) ? 0 : 4;
                const int flips = 1 + (int)(rnd() % 3);
                for (int k = 0; k < flips && len > lo; k++) b[lo + rnd() % (len - lo)] ^= (uint8_t)(1 + rnd() % 255);
            }
            rc = orc_decode_frame(b, len, dst, out_len, &out2, &used, NULL);
            if (rc == 0) n_ok++; else n_bad++;
            if (rc == 0 && (out2 > out_len || used > len)) { fprintf(stderr, "overrun reported as success\n"); return 1; }
            free(b);
        }
        free(dst);
        free(src);
    }
    printf("san_oracle ok: %llu mutated frames decoded, %llu rejected with a status\n", n_ok, n_bad);
    return 0;
}
