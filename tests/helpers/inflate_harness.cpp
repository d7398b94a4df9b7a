// Host build of the per-lane DEFLATE decoder (bonsai_amd/csrc/bns_inflate.hpp, STRIDE = 1): the CPU tier checks the decoder's logic
// against zlib without a GPU.  Test infrastructure: the product runs the same source inside inflate_members_kernel.
#define BNS_INF_FN inline
#include "../../bonsai_amd/csrc/bns_inflate.hpp"
#include <vector>
extern "C" int inf_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, uint32_t *out_n, uint32_t *crc)
{
    std::vector<uint16_t> tb(bns_inf::T_U16 + 8, 0);
    std::vector<uint8_t> scratch(bns_inf::SCRATCH_BYTES, 0);
    std::vector<uint8_t> padded(in, in + in_len);
    padded.resize(in_len + 64, 0xA5);
    bns_inf::Tables<1> t{tb.data()};
    const uint32_t st = bns_inf::inflate_member<1>(padded.data(), in_len, out, out_len, t, scratch.data(), out_n);
    uint32_t tbl[256];
    for (uint32_t i = 0; i < 256; ++i) tbl[i] = bns_inf::crc32_entry(i);
    *crc = bns_inf::crc32_bytes(tbl, out, *out_n);
    return (int)st;
}
