/* Plain-C caller of the engine's C ABI (include/stcd_hip.h): no Python, no torch, no C++.
 *   gcc -std=c99 -Iinclude examples/abi_query.c -o abi_query -Lstcd_amd -lstcd_hip -Wl,-rpath,$PWD/stcd_amd
 * Prints the parameter / BatchNorm layout a host language would mirror (names, shapes, offsets in the flat buffers) and
 * the workspace a batch needs.  Queries only: runs without a GPU (compute entry points need device pointers). */
#include <stdio.h>
#include <stdlib.h>
#include "stcd_hip.h"

int main(int argc, char** argv) {
    int arch = argc > 1 ? atoi(argv[1]) : STCD_ARCH_DIFF, batch = argc > 2 ? atoi(argv[2]) : 16, size = argc > 3 ? atoi(argv[3]) : 256;
    stcd_engine* e = NULL;
    if (stcd_create(arch, 3, 2, STCD_DTYPE_BF16, &e) != 0) { fprintf(stderr, "create: %s\n", stcd_last_error()); return 1; }
    printf("abi %d  arch %d  params %d tensors / %lld floats  bn %d layers / %lld floats\n", stcd_abi_version(), arch,
           stcd_num_params(e), (long long)stcd_param_floats(e), stcd_num_bn(e), (long long)stcd_bn_floats(e));
    for (int i = 0; i < stcd_num_params(e) && i < 4; ++i) {
        stcd_tensor_info t;
        if (stcd_param_info(e, i, &t) != 0) { fprintf(stderr, "param_info: %s\n", stcd_last_error()); return 1; }
        printf("  %-20s offset %8lld  numel %8lld  shape", t.name, (long long)t.offset, (long long)t.numel);
        for (int d = 0; d < t.ndim; ++d) printf(" %lld", (long long)t.shape[d]);
        printf("\n");
    }
    if (stcd_configure(e, batch, size, size) != 0) { fprintf(stderr, "configure: %s\n", stcd_last_error()); return 1; }
    printf("batch %d pairs of %dx%d: workspace %.1f MiB, %d dropout masks (%lld floats)\n", batch, size, size,
           stcd_workspace_bytes(e) / 1048576.0, stcd_num_dropout(e), (long long)stcd_dropout_floats(e));
    if (stcd_configure(e, 0, size, size) == 0) { fprintf(stderr, "a zero batch must be rejected\n"); return 1; }
    printf("rejected bad shape: %s\n", stcd_last_error());
    stcd_destroy(e);
    return 0;
}
