/* Plain-C caller of libzkp_hip.so (no Python, no torch): the reference's own KZG test, kzg/src/commitment.rs:36-53 --
 * SRS from secret 2, p = 1 + 2X + 3X^2, open at 1, value 6 -- through the C ABI of include/zkp_hip.h, checked with the
 * library's pairing verifier.  `c_smoke --devices N` then repeats the commitment path over N device slots (zkp_init_devices; on a
 * box with fewer GPUs the slots share device 0): the SRS is sharded at zkp_g1_bases_create, zkp_msm_g1 runs one Pippenger per
 * slot and adds the partial sums, and the result must equal the single-slot one bit for bit (the "two halves" identity of
 * SURVEY 8e behind the C ABI).  Built and run by tests/test_abi_c_caller.py (run needs a GPU).
 *   gcc -O2 -I include tests/abi/c_smoke.c -o c_smoke -L zkp-implementation_amd -lzkp_hip -Wl,-rpath,...  */
#include <stdio.h>
#include <stdlib.h>
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

/* N device slots: sharded bases against single-slot bases, plain and expanded, MSM / partial + sum / commit / open */
static int multi_device(int nslots) {
    enum { NPTS = 6000 };
    int devs[64];
    int visible = nslots; /* slots share device 0 when the box has fewer GPUs than slots */
    if (nslots > 64) nslots = 64;
    if (zkp_init_devices(NULL, nslots) != ZKP_OK) visible = 1;
    if (visible == 1) {
        for (int i = 0; i < nslots; i++) devs[i] = 0;
        CHECK(zkp_init_devices(devs, nslots));
    }
    if (zkp_device_count() != nslots) {
        fprintf(stderr, "zkp_device_count() = %d, expected %d\n", zkp_device_count(), nslots);
        return 1;
    }
    uint64_t *srs_xy = malloc(sizeof(uint64_t) * 12 * NPTS), *sc = malloc(sizeof(uint64_t) * 4 * NPTS);
    if (!srs_xy || !sc) return 1;
    CHECK(zkp_srs_g1(FR3, NPTS, srs_xy));
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 4 * NPTS; i++) { /* any four limbs below 2^254 are a valid Montgomery residue */
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        sc[i] = (i & 3) == 3 ? (x >> 3) : x;
    }
    memset(sc, 0, 32);                    /* a zero scalar */
    zkp_bases *sharded = NULL, *single = NULL;
    CHECK(zkp_g1_bases_create(srs_xy, NULL, NPTS, &sharded)); /* default thread slot: sharded over all slots */
    CHECK(zkp_set_device(nslots - 1));
    CHECK(zkp_g1_bases_create(srs_xy, NULL, NPTS, &single));  /* one slot, the last one */
    CHECK(zkp_set_device(-1));
    if (zkp_g1_bases_len(sharded) != NPTS || zkp_g1_bases_len(single) != NPTS) return 1;
    for (int pass = 0; pass < 2; pass++) {
        const size_t lens[3] = {NPTS, NPTS / 2 + 1, 5}; /* the last two leave some (or most) chunks without scalars */
        for (int k = 0; k < 3; k++) {
            uint64_t a[12], b[12], part[24], c[12];
            uint8_t ai = 0, bi = 0, ci = 0;
            CHECK(zkp_msm_g1(sharded, sc, lens[k], a, &ai));
            CHECK(zkp_msm_g1(single, sc, lens[k], b, &bi));
            CHECK(zkp_msm_g1_partial(sharded, sc, lens[k], part));
            CHECK(zkp_g1_xyzz_sum(part, 1, c, &ci));
            if (ai != bi || ai != ci || memcmp(a, b, 96) || memcmp(a, c, 96)) {
                fprintf(stderr, "sharded MSM differs from the single-slot one (pass %d, n = %zu)\n", pass, lens[k]);
                return 1;
            }
        }
        uint64_t ca[12], cb[12], oa[12], ob[12], ea[4], eb[4];
        uint8_t i1 = 0, i2 = 0, i3 = 0, i4 = 0;
        CHECK(zkp_kzg_commit(sharded, sc, NPTS, ca, &i1));
        CHECK(zkp_kzg_commit(single, sc, NPTS, cb, &i2));
        CHECK(zkp_kzg_open(sharded, sc, NPTS, FR2, oa, &i3, ea)); /* 6000 coefficients: the device path of open */
        CHECK(zkp_kzg_open(single, sc, NPTS, FR2, ob, &i4, eb));
        if (i1 != i2 || i3 != i4 || memcmp(ca, cb, 96) || memcmp(oa, ob, 96) || memcmp(ea, eb, 32)) {
            fprintf(stderr, "sharded commit / open differs (pass %d)\n", pass);
            return 1;
        }
        if (pass == 0) { /* second pass: every chunk expanded on its own slot (shared bucket set) */
            /* all-or-nothing: a refused expansion (budget of 1000 bytes) leaves EVERY chunk plain, and another width is accepted after */
            unsigned wb = 99, sl = 99;
            setenv("ZKP_SRS_EXPAND_MAX_BYTES", "1000", 1);
            if (zkp_g1_bases_precompute(sharded, 12) != ZKP_E_NOMEM) {
                fprintf(stderr, "an expansion beyond ZKP_SRS_EXPAND_MAX_BYTES was not refused with ZKP_E_NOMEM\n");
                return 1;
            }
            unsetenv("ZKP_SRS_EXPAND_MAX_BYTES");
            CHECK(zkp_g1_bases_info(sharded, &wb, &sl));
            if (wb != 0 || sl != 0) {
                fprintf(stderr, "a refused expansion left the handle expanded (%u bits, %u slices)\n", wb, sl);
                return 1;
            }
            CHECK(zkp_g1_bases_precompute(sharded, 0));
            CHECK(zkp_g1_bases_precompute(single, 0));
        }
    }
    uint64_t out[12];
    uint8_t inf = 0;
    if (zkp_msm_g1(sharded, sc, NPTS + 1, out, &inf) != ZKP_E_SIZE) { /* scheme.rs:86 */
        fprintf(stderr, "more scalars than bases was not refused\n");
        return 1;
    }
    zkp_g1_bases_destroy(sharded);
    zkp_g1_bases_destroy(single);
    /* the Fr transform over all slots (zkp_ntt_fr_sharded: four-step, exchange by peer copies inside the library) against the
     * single-slot zkp_ntt_fr (below ZKP_NTT_SHARD_MIN_LOG it stays on one device): 2^14 elements, both directions, with and without a
     * coset; a slot count that is not a power of two is refused */
    {
        enum { LOGN = 14, NN = 1 << LOGN };
        uint64_t *a = malloc(32 * NN), *b = malloc(32 * NN);
        if (!a || !b) return 1;
        for (int i = 0; i < 4 * NN; i++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            a[i] = (i & 3) == 3 ? (x >> 3) : x;
        }
        if (nslots & (nslots - 1)) {
            memcpy(b, a, 32 * NN);
            if (zkp_ntt_fr_sharded(b, LOGN, 0, NULL) != ZKP_E_ARG) {
                fprintf(stderr, "a sharded transform over %d slots (not a power of two) was not refused\n", nslots);
                return 1;
            }
        } else {
            for (int variant = 0; variant < 4; variant++) {
                const int inverse = variant & 1;
                const uint64_t *coset = (variant & 2) ? FR3 : NULL;
                uint64_t *c = malloc(32 * NN);
                if (!c) return 1;
                memcpy(b, a, 32 * NN);
                memcpy(c, a, 32 * NN);
                CHECK(zkp_ntt_fr_sharded(b, LOGN, inverse, coset));
                CHECK(zkp_ntt_fr(c, LOGN, inverse, coset));
                if (memcmp(b, c, 32 * NN)) {
                    fprintf(stderr, "sharded transform differs from the single-slot one (inverse %d, coset %d)\n", inverse, coset != NULL);
                    return 1;
                }
                free(c);
            }
            zkp_ntt_shard_geometry geo;
            CHECK(zkp_ntt_fr_sharded_geometry(LOGN, 0, 0, &geo));
            if ((int)geo.slots != nslots || geo.slab * (size_t)nslots != NN || geo.cw * geo.chunks != geo.r2) {
                fprintf(stderr, "zkp_ntt_fr_sharded_geometry is inconsistent\n");
                return 1;
            }
        }
        free(a);
        free(b);
    }
    free(srs_xy);
    free(sc);
    zkp_shutdown();
    printf("c_smoke ok: %d device slots (%s), sharded == single-slot for msm / partial / commit / open, plain and expanded, and for the Fr "
           "transform\n", nslots,
           visible == 1 ? "sharing device 0" : "one GPU each");
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 3 && strcmp(argv[1], "--devices") == 0) return multi_device(atoi(argv[2]));
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
