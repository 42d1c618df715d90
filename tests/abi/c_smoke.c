/* Plain-C caller of libzkp_hip.so (no Python, no torch): the reference's own KZG test, kzg/src/commitment.rs:36-53 --
 * SRS from secret 2, p = 1 + 2X + 3X^2, open at 1, value 6 -- through the C ABI of include/zkp_hip.h, checked with the
 * library's pairing verifier.  Built and run by tests/test_abi_c_caller.py (run needs a GPU).
 *   gcc -O2 -I include tests/abi/c_smoke.c -o c_smoke -L zkp-implementation_amd -lzkp_hip -Wl,-rpath,...  */
#include <stdio.h>
#include <string.h>
#include "zkp_hip.h"

/* Fr in memory form (Montgomery, R = 2^256) */
static const uint64_t FR1[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t FR2[4] = {0x00000003fffffffcull, 0xb1096ff400069004ull, 0x33189fdfd9789feaull, 0x304962b3598a0adfull};
static const uint64_t FR3[4] = {0x00000005fffffffaull, 0x098e27ee0009d806ull, 0xcca4efcfc634efe0ull, 0x486e140d064f104eull};
static const uint64_t FR6[4] = {0x0000000cfffffff3ull, 0xbf5eabd90015540dull, 0x6610079782c807baull, 0x1cee80c6e300a355ull};

#define CHECK(call)                                                         \
    do {                                                                    \
        int rc_ = (call);                                                   \
        if (rc_ != ZKP_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zkp_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(void) {
    CHECK(zkp_init(-1));
    uint64_t srs_xy[13 * 12];
    CHECK(zkp_srs_g1(FR2, 13, srs_xy)); /* Srs::new_from_secret(2, 10): 10 + 3 points */
    zkp_bases *bases = NULL;
    CHECK(zkp_g1_bases_create(srs_xy, NULL, 13, &bases));
    uint64_t coeffs[12];
    memcpy(coeffs, FR1, 32);
    memcpy(coeffs + 4, FR2, 32);
    memcpy(coeffs + 8, FR3, 32);
    uint64_t commit[12], opening[12], eval[4];
    uint8_t cinf = 0, oinf = 0;
    CHECK(zkp_kzg_commit(bases, coeffs, 3, commit, &cinf));
    CHECK(zkp_kzg_open(bases, coeffs, 3, FR1, opening, &oinf, eval));
    if (memcmp(eval, FR6, 32) != 0) {
        fprintf(stderr, "p(1) != 6\n");
        return 1;
    }
    uint64_t g2[24], g2s[24];
    uint8_t ginf = 0;
    CHECK(zkp_g2_generator(g2));
    CHECK(zkp_g2_mul(g2, 0, FR2, g2s, &ginf));
    int ok = 0, bad = 1;
    CHECK(zkp_kzg_verify(g2s, commit, cinf, opening, oinf, eval, FR1, &ok));
    CHECK(zkp_kzg_verify(g2s, commit, cinf, opening, oinf, FR3, FR1, &bad)); /* wrong value must be rejected */
    zkp_g1_bases_destroy(bases);
    zkp_shutdown();
    if (!ok || bad) {
        fprintf(stderr, "verify: accepted=%d, wrong value accepted=%d\n", ok, bad);
        return 1;
    }
    printf("c_smoke ok: commit, open and pairing check of p = 1 + 2X + 3X^2 over SRS(2)\n");
    return 0;
}
