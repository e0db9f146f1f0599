// (filled in below) QAP quotient pipeline -- placeholder until the NTT pipeline lands.
#pragma once
#include "field.cuh"
namespace ps {
struct QuotientCache;
static inline void quotient_cache_free(QuotientCache*) {}
}  // namespace ps
