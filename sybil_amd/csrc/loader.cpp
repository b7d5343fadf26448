// loader.cpp -- native reader for sybil table directories (gob column files).
// Placeholder until the gob decoder lands: fails loudly instead of pretending.
#include "engine.h"

using namespace sybl;

extern "C" int sybl_table_open(sybl_ctx *ctx, const char *dir, const char *table, const char *const *columns,
                               int32_t n_columns, int32_t rank, int32_t nranks, sybl_table **out) {
    (void)ctx; (void)dir; (void)table; (void)columns; (void)n_columns; (void)rank; (void)nranks;
    if (out) *out = nullptr;
    return fail(SYBL_E_IO, "sybl_table_open: the gob table loader is not built into this version");
}
