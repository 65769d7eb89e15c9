// Internals shared between the translation units of the device layer.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lpt.h"

int lpt_internal_device(const lpt_ctx* ctx);
hipStream_t lpt_internal_stream(const lpt_ctx* ctx);
// Records `msg` as the context's last error and returns `code`.
int lpt_internal_fail(lpt_ctx* ctx, int code, const char* msg);
