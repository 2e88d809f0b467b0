// placeholder while the index kernels are being written
#include "kmi_internal.h"
using namespace kmi;
extern "C" {
kmi_status kmi_route_dev(kmi_ctx *ctx, const kmi_config *, const uint64_t *, size_t, uint32_t, uint64_t *, uint64_t *) { return set_err(ctx, KMI_ERR_INVALID, "not implemented"); }
kmi_status kmi_index_create(kmi_ctx *ctx, const kmi_config *, kmi_index **) { return set_err(ctx, KMI_ERR_INVALID, "not implemented"); }
kmi_status kmi_index_destroy(kmi_index *) { return KMI_ERR_INVALID; }
kmi_status kmi_index_insert_host(kmi_index *, const uint64_t *, size_t) { return KMI_ERR_INVALID; }
kmi_status kmi_index_insert_dev(kmi_index *, const uint64_t *, size_t) { return KMI_ERR_INVALID; }
kmi_status kmi_index_build_host(kmi_index *, const uint8_t *, size_t, uint64_t) { return KMI_ERR_INVALID; }
kmi_status kmi_index_build_dev(kmi_index *, const uint8_t *, size_t, uint64_t) { return KMI_ERR_INVALID; }
kmi_status kmi_index_local_size(kmi_index *, uint64_t *) { return KMI_ERR_INVALID; }
kmi_status kmi_index_export_host(kmi_index *, uint64_t *, uint32_t *, size_t, uint64_t *) { return KMI_ERR_INVALID; }
void kmi_results_free(kmi_results *r) { if (r) { free(r->keys); free(r->values); memset(r, 0, sizeof(*r)); } }
kmi_status kmi_index_count_host(kmi_index *, const uint64_t *, size_t, kmi_results *) { return KMI_ERR_INVALID; }
kmi_status kmi_index_find_host(kmi_index *, const uint64_t *, size_t, kmi_results *) { return KMI_ERR_INVALID; }
kmi_status kmi_index_erase_host(kmi_index *, const uint64_t *, size_t, uint64_t *) { return KMI_ERR_INVALID; }
kmi_status kmi_index_count_dev(kmi_index *, const uint64_t *, size_t, uint64_t *, uint64_t *, uint64_t *) { return KMI_ERR_INVALID; }
kmi_status kmi_index_find_dev(kmi_index *, const uint64_t *, size_t, uint64_t *, uint64_t *, uint64_t *) { return KMI_ERR_INVALID; }
}
