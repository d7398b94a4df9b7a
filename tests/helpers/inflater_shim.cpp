// A CPU stand-in for the bns_inflater_* entry points (include/bonsai_amd.h), LD_PRELOADed in front of libbonsai_amd.so by
// tests/test_inflate.py: the host reader's GPU-inflate threads (batches from the back of the task queue, staging, windows) run
// against it in the CPU tier.  It decodes with the product's own decoder source compiled for the host (bns_inflate.hpp, STRIDE = 1).
// Test infrastructure only.
#define BNS_INF_FN inline
#include "../../bonsai_amd/csrc/bns_inflate.hpp"
#include <cstdlib>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
struct bns_inflater { std::string err; };
extern "C" {
int bns_inflater_create(int device, bns_inflater **out) { if (device != 0 || !out) return -1; *out = new bns_inflater; return 0; }
void bns_inflater_destroy(bns_inflater *h) { delete h; }
const char *bns_inflater_error(const bns_inflater *h) { return h ? h->err.c_str() : "null"; }
float bns_inflater_last_kernel_ms(const bns_inflater *) { return 1.f; }
int bns_inflater_host_alloc(bns_inflater *, size_t n, void **out) { *out = std::malloc(n ? n : 4); return *out ? 0 : -2; }
int bns_inflater_host_free(bns_inflater *, void *p) { std::free(p); return 0; }
int bns_inflate_members(bns_inflater *, const uint8_t *comp, uint64_t, const uint64_t *in_off, const uint32_t *in_len, const uint64_t *out_off,
                        const uint32_t *out_len, uint64_t n, uint8_t *text, uint64_t, uint32_t *crc32, uint32_t *status)
{
    std::vector<uint16_t> tb(bns_inf::T_U16 + 8);
    std::vector<uint8_t> scratch(bns_inf::SCRATCH_BYTES);
    uint32_t tbl[256];
    for (uint32_t i = 0; i < 256; ++i) tbl[i] = bns_inf::crc32_entry(i);
    if (const char *e = std::getenv("BNS_SHIM_LATENCY_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(e)));   // (the device answers late)
    for (uint64_t m = 0; m < n; ++m) {
        bns_inf::Tables<1> t{tb.data()};
        uint32_t got = 0;
        status[m] = bns_inf::inflate_member<1>(comp + in_off[m], in_len[m], text + out_off[m], out_len[m], t, scratch.data(), &got);
        crc32[m] = bns_inf::crc32_bytes(tbl, text + out_off[m], got);
    }
    return 0;
}
}
