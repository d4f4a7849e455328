// Rows [first, end) of an image a launch covers (the resolved form of gr_rows).
#pragma once
#include <stdint.h>

struct RowSpan
{
	uint32_t first, end;
	uint32_t count() const { return end - first; }
};
