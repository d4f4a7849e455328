// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
// GTX ("GRANITE TEXFMT1"): the container Granite stores textures and image dumps in
// (vulkan/texture/memory_mapped_texture.cpp:29-44 header, vulkan/texture/texture_format.cpp:349-387 payload layout).
// The wire format on either side of the image-space chain: G-buffer attachments dumped by a Granite build come in as
// .gtx, the frames this executor produces go out as .gtx (tools/image_compare-style checks on the Vulkan side).
//
// Header, 64 bytes, little endian: magic[16] = "GRANITE TEXFMT1\0"; u32 type (VkImageType), format (VkFormat), width,
// height, depth, layers, levels, flags; u64 payload_size; u64 reserved.  Payload: mip levels in order, each starting at
// a 16-byte aligned offset; inside a level the array layers (and depth slices) follow each other, rows tightly packed.
// Only uncompressed formats the executor knows (vk_format_block_size) are accepted.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include "vk_subset.hpp"

namespace Granite
{
struct GtxImage
{
	uint32_t type = 1; // VK_IMAGE_TYPE_2D
	VkFormat format = VK_FORMAT_UNDEFINED;
	uint32_t width = 0, height = 0, depth = 1, layers = 1, levels = 1;
	uint32_t flags = 0; // MemoryMappedTextureFlags: cube / mipgen-on-load bits and the component swizzle, kept verbatim
	std::vector<uint8_t> payload;

	static constexpr size_t HeaderSize = 64;
	size_t level_offset(uint32_t level) const; // into payload
	size_t level_size(uint32_t level) const;   // all layers of that level
	size_t required_payload_size() const;
	uint32_t level_width(uint32_t level) const { return (width >> level) ? (width >> level) : 1u; }
	uint32_t level_height(uint32_t level) const { return (height >> level) ? (height >> level) : 1u; }
};

// Throws std::runtime_error with the reason (bad magic, unsupported format, truncated / oversized payload ...).
GtxImage gtx_parse(const void *data, size_t size);
GtxImage gtx_load(const std::string &path);
std::vector<uint8_t> gtx_serialize(const GtxImage &image);
void gtx_save(const GtxImage &image, const std::string &path);
} // namespace Granite
