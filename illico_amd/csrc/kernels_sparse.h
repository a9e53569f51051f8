// CSC / CSR ingest kernels (filled in below the dense path).
#pragma once
#include "common.h"
