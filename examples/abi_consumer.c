/* A plain-C consumer of the libadm boundary: what a binding in ANY language does first.  Compiled as C99 against include/adm.h
 * and linked to adorym_amd/libadm.so by tests/test_abi_c_consumer.py (no Python between this file and the library).
 *   gcc -std=c99 -Wall -Werror -Iinclude examples/abi_consumer.c -o /tmp/abi_consumer -Ladorym_amd -ladm -Wl,-rpath,$PWD/adorym_amd
 * Without a GPU it checks the version, the device count and the error contract (negative status + message, no crash); with one
 * (argv[1] = "gpu") it runs the smallest device round trip: context, allocation, memset, copies, event timing.            */
#include <stdio.h>
#include <string.h>
#include "adm.h"

int main(int argc, char** argv) {
    if (adm_version() != ADM_VERSION) { printf("version mismatch: %d\n", adm_version()); return 1; }
    const int n_dev = adm_device_count();
    printf("adm_version %d, devices %d\n", adm_version(), n_dev);
    adm_ctx* ctx = NULL;
    if (argc < 2 || strcmp(argv[1], "gpu") != 0) {
        /* error contract: a bad call returns a negative adm_status and leaves a message; nothing is created */
        int rc = adm_ctx_create(n_dev + 7, NULL, &ctx);
        if (rc >= 0 || ctx != NULL) { printf("expected a failure for device %d, got %d\n", n_dev + 7, rc); return 2; }
        const char* msg = adm_last_error();
        if (!msg || !msg[0]) { printf("no error message\n"); return 3; }
        printf("adm_ctx_create(device %d) -> %d: %s\n", n_dev + 7, rc, msg);
        if (adm_ctx_sync(NULL) != ADM_ERR_INVALID) { printf("adm_ctx_sync(NULL) should be ADM_ERR_INVALID\n"); return 4; }
        printf("OK (no GPU needed)\n");
        return 0;
    }
    if (adm_ctx_create(0, NULL, &ctx) != ADM_OK) { printf("adm_ctx_create: %s\n", adm_last_error()); return 5; }
    enum { N = 1 << 20 };
    static float host[N], back[N];
    for (int i = 0; i < N; ++i) host[i] = (float)i * 0.5f;
    void *a = NULL, *b = NULL, *e0 = NULL, *e1 = NULL;
    float ms = -1.f;
    int rc = adm_malloc(ctx, sizeof host, &a);
    if (!rc) rc = adm_malloc(ctx, sizeof host, &b);
    if (!rc) rc = adm_event_create(ctx, &e0);
    if (!rc) rc = adm_event_create(ctx, &e1);
    if (!rc) rc = adm_h2d(ctx, a, host, sizeof host);
    if (!rc) rc = adm_memset(ctx, b, 0, sizeof host);
    if (!rc) rc = adm_event_record(ctx, e0);
    if (!rc) rc = adm_d2d(ctx, b, a, sizeof host);
    if (!rc) rc = adm_event_record(ctx, e1);
    if (!rc) rc = adm_d2h(ctx, back, b, sizeof host);
    if (!rc) rc = adm_event_elapsed_ms(ctx, e0, e1, &ms);
    if (rc) { printf("device round trip failed (%d): %s\n", rc, adm_last_error()); return 6; }
    if (memcmp(host, back, sizeof host) != 0) { printf("round trip changed the data\n"); return 7; }
    printf("device %d: 4 MiB round trip bit-exact, device copy %.3f ms\n", adm_ctx_device(ctx), ms);
    adm_event_destroy(ctx, e0); adm_event_destroy(ctx, e1);
    adm_free(ctx, a); adm_free(ctx, b);
    adm_ctx_destroy(ctx);
    printf("OK (gpu)\n");
    return 0;
}
