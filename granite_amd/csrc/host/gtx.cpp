// Follows MIT-licensed work (Granite, (c) 2017-2026 Hans-Kristian Arntzen; FidelityFX parts (c) 2021 Advanced Micro Devices, Inc.): see
// THIRD_PARTY_NOTICES.md at the repository root.
#include "gtx.hpp"
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace Granite
{
static const char GtxMagic[16] = "GRANITE TEXFMT1";

size_t GtxImage::level_size(uint32_t level) const
{
	const uint32_t d = (depth >> level) ? (depth >> level) : 1u;
	return size_t(level_width(level)) * level_height(level) * d * layers * vk_format_block_size(format);
}

size_t GtxImage::level_offset(uint32_t level) const
{
	size_t offset = 0;
	for (uint32_t l = 0;; l++)
	{
		offset = (offset + 15) & ~size_t(15);
		if (l == level)
			return offset;
		offset += level_size(l);
	}
}

size_t GtxImage::required_payload_size() const
{
	return levels ? level_offset(levels - 1) + level_size(levels - 1) : 0;
}

static uint32_t get32(const uint8_t *p)
{
	return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24);
}

static uint64_t get64(const uint8_t *p)
{
	return uint64_t(get32(p)) | (uint64_t(get32(p + 4)) << 32);
}

static void put32(uint8_t *p, uint32_t v)
{
	for (int i = 0; i < 4; i++)
		p[i] = uint8_t(v >> (8 * i));
}

static void put64(uint8_t *p, uint64_t v)
{
	put32(p, uint32_t(v));
	put32(p + 4, uint32_t(v >> 32));
}

GtxImage gtx_parse(const void *data, size_t size)
{
	auto *bytes = static_cast<const uint8_t *>(data);
	if (size < GtxImage::HeaderSize || memcmp(bytes, GtxMagic, sizeof(GtxMagic)) != 0)
		throw std::runtime_error("Not a GTX file (magic \"GRANITE TEXFMT1\" missing).");

	GtxImage img;
	img.type = get32(bytes + 16);
	img.format = VkFormat(get32(bytes + 20));
	img.width = get32(bytes + 24);
	img.height = get32(bytes + 28);
	img.depth = get32(bytes + 32);
	img.layers = get32(bytes + 36);
	img.levels = get32(bytes + 40);
	img.flags = get32(bytes + 44);
	const uint64_t payload_size = get64(bytes + 48);

	if (img.type > 2) // VK_IMAGE_TYPE_1D / 2D / 3D
		throw std::runtime_error("GTX: unknown image type.");
	if (!vk_format_block_size(img.format))
		throw std::runtime_error("GTX: format " + std::to_string(unsigned(img.format)) + " is not one the executor handles.");
	if (!img.width || !img.height || !img.depth || !img.layers || !img.levels || img.levels > 16)
		throw std::runtime_error("GTX: empty or implausible dimensions.");
	// Bound every extent before any size arithmetic: width * height * depth * layers * bpp must not wrap size_t (a header
	// declaring 65536 x 65536 x 32768 x 32768 layers would otherwise "need" 0 bytes and pass the payload check below).
	if (img.width > 65536u || img.height > 65536u || img.depth > 65536u || img.layers > 65536u ||
	    uint64_t(img.depth) * img.layers > (1u << 20))
		throw std::runtime_error("GTX: empty or implausible dimensions.");
	if (payload_size > size - GtxImage::HeaderSize)
		throw std::runtime_error("GTX: file is truncated.");
	// memory_mapped_texture.cpp:318-321: the header must describe exactly the payload, and the file must hold it.
	if (payload_size != img.required_payload_size())
		throw std::runtime_error("GTX: payload size does not match the header's layout.");
	if (size < GtxImage::HeaderSize + payload_size)
		throw std::runtime_error("GTX: file is truncated.");
	img.payload.assign(bytes + GtxImage::HeaderSize, bytes + GtxImage::HeaderSize + payload_size);
	return img;
}

std::vector<uint8_t> gtx_serialize(const GtxImage &image)
{
	if (!vk_format_block_size(image.format))
		throw std::runtime_error("GTX: cannot serialise this format.");
	const size_t payload_size = image.required_payload_size();
	if (image.payload.size() != payload_size)
		throw std::runtime_error("GTX: payload does not match the layout (mip levels start at 16-byte aligned offsets).");
	std::vector<uint8_t> out(GtxImage::HeaderSize + payload_size, 0);
	memcpy(out.data(), GtxMagic, sizeof(GtxMagic));
	put32(out.data() + 16, image.type);
	put32(out.data() + 20, uint32_t(image.format));
	put32(out.data() + 24, image.width);
	put32(out.data() + 28, image.height);
	put32(out.data() + 32, image.depth);
	put32(out.data() + 36, image.layers);
	put32(out.data() + 40, image.levels);
	put32(out.data() + 44, image.flags);
	put64(out.data() + 48, payload_size);
	put64(out.data() + 56, 0);
	memcpy(out.data() + GtxImage::HeaderSize, image.payload.data(), payload_size);
	return out;
}

GtxImage gtx_load(const std::string &path)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f)
		throw std::runtime_error("GTX: cannot open " + path);
	std::vector<uint8_t> raw;
	uint8_t chunk[1 << 16];
	size_t n;
	while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0)
		raw.insert(raw.end(), chunk, chunk + n);
	fclose(f);
	try
	{
		return gtx_parse(raw.data(), raw.size());
	}
	catch (const std::runtime_error &e)
	{
		throw std::runtime_error(path + ": " + e.what());
	}
}

void gtx_save(const GtxImage &image, const std::string &path)
{
	auto raw = gtx_serialize(image);
	FILE *f = fopen(path.c_str(), "wb");
	if (!f)
		throw std::runtime_error("GTX: cannot create " + path);
	const size_t written = fwrite(raw.data(), 1, raw.size(), f);
	if (fclose(f) != 0 || written != raw.size())
		throw std::runtime_error("GTX: short write to " + path);
}
} // namespace Granite
