/* CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 -- the checksum of TensorFlow's tensor
 * bundle files (tensorflow/core/lib/hash/crc32c.h: every table block and every tensor's bytes carry a masked
 * crc32c).  Host-only helper of hpmn_amd/tf_checkpoint.py: a 200 MB embedding table is checksummed in ~0.2 s
 * here and in minutes by a Python loop.  Build: gcc -O2 -shared -fPIC (hpmn_amd/build.py). */
#include <stddef.h>
#include <stdint.h>

static uint32_t T[8][256];
static int ready = 0;

static void init(void) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
        T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int s = 1; s < 8; ++s) T[s][i] = (T[s - 1][i] >> 8) ^ T[0][T[s - 1][i] & 0xffu];
    ready = 1;
}

/* crc of (previous bytes || data): pass 0 for a fresh checksum (pre/post inversion is handled here) */
uint32_t hpmn_crc32c_extend(uint32_t crc, const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    if (!ready) init();
    uint32_t c = crc ^ 0xffffffffu;
    while (n && ((uintptr_t)p & 7u)) { c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu]; --n; }
    while (n >= 8) {
        const uint32_t lo = *(const uint32_t *)p ^ c, hi = *(const uint32_t *)(p + 4);
        c = T[7][lo & 0xffu] ^ T[6][(lo >> 8) & 0xffu] ^ T[5][(lo >> 16) & 0xffu] ^ T[4][lo >> 24] ^
            T[3][hi & 0xffu] ^ T[2][(hi >> 8) & 0xffu] ^ T[1][(hi >> 16) & 0xffu] ^ T[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xffu];
    return c ^ 0xffffffffu;
}
