/*
 * ref_harness.cpp -- thin C exports around the REFERENCE's own headers, compiled where they lie
 * under $(REF) (default /root/reference) into oracle/_ref/libbns_ref.so.  Test infrastructure only.
 *
 * Only reference headers that compile with no stand-ins are used:
 *   include/bonsai/khash64.h  (kh_init/put/get/resize, __ac_Wang64_hash, flag macros)
 *   linear/linear.h           (linear::counter, linear::set)
 * The instantiations below repeat include/bonsai/util.h:160,162 (khash_t(c), khash_t(p)) because
 * util.h itself needs the un-vendored sketch/zlib-ng/ntHash submodules and is unbuildable here.
 * No reference source is copied; this file only calls into the headers.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include "include/bonsai/khash64.h"
#include "linear/linear.h"

typedef uint32_t tax_t;
KHASH_MAP_INIT_INT64(c, tax_t)   /* util.h:160 */
KHASH_MAP_INIT_INT(p, tax_t)     /* util.h:162 */

extern "C" {

uint64_t ref_wang64(uint64_t k) { return __ac_Wang64_hash(k); }

void *ref_khc_new(void) { return kh_init(c); }
void ref_khc_free(void *h) { kh_destroy(c, (khash_t(c) *)h); }
/* sequential insert exactly as feature_min.h:212-216 does (kh_get, then kh_put + assign on miss;
 * on hit the value is overwritten with `val` here -- callers pass unique keys when layout matters) */
void ref_khc_insert(void *hv, const uint64_t *keys, const uint32_t *vals, uint64_t n)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    int khr;
    for (uint64_t i = 0; i < n; ++i) {
        khint_t k2 = kh_get(c, h, keys[i]);
        if (k2 == kh_end(h)) k2 = kh_put(c, h, keys[i], &khr);
        kh_val(h, k2) = vals[i];
    }
}
int ref_khc_resize(void *hv, uint64_t nb) { return kh_resize(c, (khash_t(c) *)hv, nb); }
void ref_khc_del_key(void *hv, uint64_t key)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    khint_t k2 = kh_get(c, h, key);
    if (k2 != kh_end(h)) kh_del(c, h, k2);
}
void ref_khc_info(void *hv, uint64_t *out4, uint32_t **flags, uint64_t **keys, uint32_t **vals)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    out4[0] = h->n_buckets; out4[1] = h->size; out4[2] = h->n_occupied; out4[3] = h->upper_bound;
    *flags = h->flags; *keys = (uint64_t *)h->keys; *vals = h->vals;
}
void ref_khc_get_batch(void *hv, const uint64_t *keys, uint64_t n, uint32_t *out_val, uint8_t *out_found)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    for (uint64_t i = 0; i < n; ++i) {
        khint_t k2 = kh_get(c, h, keys[i]);
        out_found[i] = (k2 != kh_end(h));
        out_val[i] = out_found[i] ? kh_val(h, k2) : 0;
    }
}
/* raw slot index returned by kh_get (kh_end == n_buckets on miss) */
uint64_t ref_khc_get(void *hv, uint64_t key) { return kh_get(c, (khash_t(c) *)hv, key); }

/* linear::counter<tax_t,u16> as classifier.h:10 declares it */
uint32_t ref_counter(const uint32_t *adds, uint32_t n, uint32_t *keys_out, uint16_t *vals_out)
{
    linear::counter<tax_t, uint16_t> ct;
    for (uint32_t i = 0; i < n; ++i) ct.add(adds[i]);
    for (uint32_t i = 0; i < ct.size(); ++i) { keys_out[i] = ct.keys()[i]; vals_out[i] = ct.vals()[i]; }
    return ct.size();
}
uint16_t ref_counter_count(const uint32_t *adds, uint32_t n, uint32_t key)
{
    linear::counter<tax_t, uint16_t> ct;
    for (uint32_t i = 0; i < n; ++i) ct.add(adds[i]);
    return ct.count(key);
}
/* linear::set insertion order (used by resolve_tree's max_taxa and lca's `nodes`) */
uint32_t ref_linear_set(const uint32_t *ins, uint32_t n, uint32_t *out)
{
    linear::set<tax_t> s;
    for (uint32_t i = 0; i < n; ++i) s.insert(ins[i]);
    uint32_t m = 0;
    for (auto v : s) out[m++] = v;
    return m;
}

} /* extern "C" */
