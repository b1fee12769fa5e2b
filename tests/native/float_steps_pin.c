/* Pins send-slam_amd/csrc/ss_float_steps.h (host build of the code the device runs):
 *   ss_sincosf     against this container's glibc sinf/cosf, every float in [2^-15, 120)
 *                  (stride from argv[1], default 1)
 *   ss_fast_atan2  against the oracle's orc_fast_atan2 on integer moment pairs
 * Test infrastructure.  Exit code 0 = no mismatch. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../send-slam_amd/csrc/ss_float_steps.h"
#include "../../oracle/orb_oracle.h"

int main(int argc, char **argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
    long n = 0, ms = 0, mc = 0, ma = 0;
    for (uint32_t bits = 0x38000000u; bits < 0x42F00000u; bits += stride) {
        float y, s, c;
        memcpy(&y, &bits, 4);
        ss_sincosf(y, &s, &c);
        n++;
        ms += (s != sinf(y));
        mc += (c != cosf(y));
    }
    float s0, c0;
    ss_sincosf(0.0f, &s0, &c0);
    ms += (s0 != sinf(0.0f));
    mc += (c0 != cosf(0.0f));
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < 4000000; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const float m01 = (float)((int32_t)(st & 0x3FFFFF) - 0x200000);
        const float m10 = (float)((int32_t)((st >> 32) & 0x3FFFFF) - 0x200000);
        float a = ss_fast_atan2(m01, m10), b = orc_fast_atan2(m01, m10);
        ma += (memcmp(&a, &b, 4) != 0);
    }
    printf("n=%ld sin_mismatch=%ld cos_mismatch=%ld atan2_mismatch=%ld\n", n, ms, mc, ma);
    return (ms || mc || ma) ? 1 : 0;
}
