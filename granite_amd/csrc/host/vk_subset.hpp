// The handful of Vulkan enumerants Granite's RenderGraph declarations mention (renderer/render_graph.hpp:124-251,
// renderer/post/hdr.cpp:313-353, ...), with Vulkan's numeric values, so pass-declaration code reads like the
// reference's without any Vulkan header.  Nothing here talks to a Vulkan driver.
#pragma once
#include <cstdint>

using VkFlags = uint32_t;
using VkFlags64 = uint64_t;
using VkDeviceSize = uint64_t;
using VkImageUsageFlags = VkFlags;
using VkBufferUsageFlags = VkFlags;
using VkPipelineStageFlags2 = VkFlags64;
using VkAccessFlags2 = VkFlags64;

enum VkFormat : uint32_t
{
	VK_FORMAT_UNDEFINED = 0,
	VK_FORMAT_R8_UNORM = 9,
	VK_FORMAT_R8G8_UNORM = 16,
	VK_FORMAT_R8G8B8A8_UNORM = 37,
	VK_FORMAT_R8G8B8A8_SRGB = 43,
	VK_FORMAT_A2B10G10R10_UNORM_PACK32 = 64,
	VK_FORMAT_R16_SFLOAT = 76,
	VK_FORMAT_R16G16_SFLOAT = 83,
	VK_FORMAT_R16G16B16A16_SFLOAT = 97,
	VK_FORMAT_R32_SFLOAT = 100,
	VK_FORMAT_B10G11R11_UFLOAT_PACK32 = 122,
	VK_FORMAT_D16_UNORM = 124,
	VK_FORMAT_D32_SFLOAT = 126
};

enum VkImageUsageFlagBits : uint32_t
{
	VK_IMAGE_USAGE_TRANSFER_SRC_BIT = 0x1,
	VK_IMAGE_USAGE_TRANSFER_DST_BIT = 0x2,
	VK_IMAGE_USAGE_SAMPLED_BIT = 0x4,
	VK_IMAGE_USAGE_STORAGE_BIT = 0x8,
	VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT = 0x10,
	VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT = 0x20,
	VK_IMAGE_USAGE_INPUT_ATTACHMENT_BIT = 0x80
};

enum VkBufferUsageFlagBits : uint32_t
{
	VK_BUFFER_USAGE_TRANSFER_SRC_BIT = 0x1,
	VK_BUFFER_USAGE_TRANSFER_DST_BIT = 0x2,
	VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT = 0x10,
	VK_BUFFER_USAGE_STORAGE_BUFFER_BIT = 0x20,
	VK_BUFFER_USAGE_INDEX_BUFFER_BIT = 0x40,
	VK_BUFFER_USAGE_VERTEX_BUFFER_BIT = 0x80,
	VK_BUFFER_USAGE_INDIRECT_BUFFER_BIT = 0x100
};

// Stage/access masks are accepted and recorded for API compatibility; a HIP stream is in-order, so they never
// turn into barriers.
constexpr VkPipelineStageFlags2 VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT = 0x800ull;
constexpr VkPipelineStageFlags2 VK_PIPELINE_STAGE_FRAGMENT_SHADER_BIT = 0x80ull;
constexpr VkPipelineStageFlags2 VK_PIPELINE_STAGE_2_COPY_BIT = 0x100000000ull;
constexpr VkAccessFlags2 VK_ACCESS_2_SHADER_STORAGE_WRITE_BIT = 0x400000000ull;
constexpr VkAccessFlags2 VK_ACCESS_2_SHADER_STORAGE_READ_BIT = 0x200000000ull;
constexpr VkAccessFlags2 VK_ACCESS_2_SHADER_SAMPLED_READ_BIT = 0x100000000ull;
constexpr VkAccessFlags2 VK_ACCESS_UNIFORM_READ_BIT = 0x8ull;
constexpr VkAccessFlags2 VK_ACCESS_TRANSFER_WRITE_BIT = 0x1000ull;

union VkClearColorValue
{
	float float32[4];
	int32_t int32[4];
	uint32_t uint32[4];
};

struct VkClearDepthStencilValue
{
	float depth;
	uint32_t stencil;
};

static inline unsigned vk_format_block_size(VkFormat format)
{
	switch (format)
	{
	case VK_FORMAT_R8_UNORM: return 1;
	case VK_FORMAT_R8G8_UNORM: return 2;
	case VK_FORMAT_D16_UNORM: return 2;
	case VK_FORMAT_R16_SFLOAT: return 2;
	case VK_FORMAT_R8G8B8A8_UNORM:
	case VK_FORMAT_R8G8B8A8_SRGB:
	case VK_FORMAT_A2B10G10R10_UNORM_PACK32:
	case VK_FORMAT_R16G16_SFLOAT:
	case VK_FORMAT_B10G11R11_UFLOAT_PACK32:
	case VK_FORMAT_R32_SFLOAT:
	case VK_FORMAT_D32_SFLOAT: return 4;
	case VK_FORMAT_R16G16B16A16_SFLOAT: return 8;
	default: return 0;
	}
}

static inline bool vk_format_is_srgb(VkFormat format) { return format == VK_FORMAT_R8G8B8A8_SRGB; }
static inline bool vk_format_has_depth(VkFormat format) { return format == VK_FORMAT_D32_SFLOAT || format == VK_FORMAT_D16_UNORM; }
