"""Helpers for the GPU lighting tests: device-side copies of an oracle-packed light scene."""
import ctypes as C

import numpy as np

from granite_amd import capi, synth
from oracle import oracle as orc


class Scene:
    def __init__(self, w, h, num_lights, res=(128, 64, 4096), seed=synth.SEED, spot_fraction=0.25, scene="default"):
        self.w, self.h = w, h
        self.cam = synth.Camera(w, h)
        self.rp = self.cam.render_params()
        self.gbuf = synth.make_gbuffer(self.cam, seed, scene=scene)
        self.descs = synth.make_lights(self.cam, num_lights, spot_fraction, seed=seed, scene=scene)
        self.res = res
        self.n, self.lights, self.model, self.type_mask, self.order = orc.pack_lights(self.descs, self.rp[99:102])
        self.prm = orc.cluster_params(self.rp, res[0], res[1], res[2], self.n)

    def cluster_params_struct(self):
        s = capi.ClusterParams()
        C.memmove(C.byref(s), self.prm.ctypes.data, 176)
        return s

    def upload_transforms(self, gr):
        buf = capi.DeviceBuffer(gr, capi.TRANSFORMS_SIZE)
        buf.upload(self.lights.view(np.uint8), capi.TRANSFORMS_OFFSET_LIGHTS)
        buf.upload(self.model.view(np.uint8), capi.TRANSFORMS_OFFSET_MODEL)
        buf.upload(self.type_mask.view(np.uint8), capi.TRANSFORMS_OFFSET_TYPE_MASK)
        return buf

    def build_clusters_gpu(self, gr):
        lib, h = gr.lib, gr.handle
        prm = self.cluster_params_struct()
        transforms = self.upload_transforms(gr)
        spots = capi.DeviceBuffer(gr, capi.TRANSFORMED_SPOT_BYTES_PER_LIGHT * 4096)
        setup = capi.DeviceBuffer(gr, capi.CULL_SETUP_BYTES_PER_LIGHT * 4096)
        bitmask = capi.DeviceBuffer(gr, self.res[0] * self.res[1] * 512)
        ranges = capi.DeviceBuffer(gr, self.res[2] * 8)
        push = capi.PushSpotTransform()
        push.vp[:] = self.rp[32:48]
        push.camera_pos[:] = self.rp[96:99]
        push.num_lights = self.n
        push.camera_front[:] = self.rp[99:102]
        push.z_near, push.z_far = self.rp[102], self.rp[103]
        gr.check(lib.gr_cluster_spot_transform(h, None, transforms.ptr, spots.ptr, push))
        ps = capi.PushClusterSetup()
        ps.view[:] = self.rp[16:32]
        ps.num_lights = self.n
        gr.check(lib.gr_cluster_setup(h, None, transforms.ptr, spots.ptr, setup.ptr, prm, ps))
        gr.check(lib.gr_cluster_binning(h, None, transforms.ptr, setup.ptr, bitmask.ptr, prm))
        zr = orc.light_z_ranges(self.rp, self.lights, self.model, self.type_mask, self.n, self.res[2])
        zr_buf = capi.DeviceBuffer(gr, zr.nbytes).upload(zr)
        pz = capi.PushZRange(len(zr), (len(zr) + 127) // 128, self.res[2])
        gr.check(lib.gr_cluster_z_range(h, None, zr_buf.ptr, ranges.ptr, pz))
        gr.sync()
        return {"transforms": transforms, "spots": spots, "setup": setup, "bitmask": bitmask, "range": ranges, "zr": zr}

    def build_clusters_gpu_fused(self, gr, pinned: bool):
        """The same build as two launches: gr_cluster_front (uploads + spot_transform + setup + z_range) and gr_cluster_binning.
        pinned: the CPU-packed arrays are read from pinned host memory by the launch itself (what the executor does), else they
        are uploaded into the transforms buffer first and the launch takes them from there."""
        lib, h = gr.lib, gr.handle
        prm = self.cluster_params_struct()
        transforms = capi.DeviceBuffer(gr, capi.TRANSFORMS_SIZE) if pinned else self.upload_transforms(gr)
        spots = capi.DeviceBuffer(gr, capi.TRANSFORMED_SPOT_BYTES_PER_LIGHT * 4096)
        setup = capi.DeviceBuffer(gr, capi.CULL_SETUP_BYTES_PER_LIGHT * 4096)
        bitmask = capi.DeviceBuffer(gr, self.res[0] * self.res[1] * 512)
        ranges = capi.DeviceBuffer(gr, self.res[2] * 8)
        zr = orc.light_z_ranges(self.rp, self.lights, self.model, self.type_mask, self.n, self.res[2])
        zr_buf = capi.DeviceBuffer(gr, zr.nbytes)
        push = capi.PushSpotTransform()
        push.vp[:] = self.rp[32:48]
        push.camera_pos[:] = self.rp[96:99]
        push.num_lights = self.n
        push.camera_front[:] = self.rp[99:102]
        push.z_near, push.z_far = self.rp[102], self.rp[103]
        ps = capi.PushClusterSetup()
        ps.view[:] = self.rp[16:32]
        ps.num_lights = self.n
        pz = capi.PushZRange(len(zr), (len(zr) + 127) // 128, self.res[2])
        a = capi.ClusterFrontArgs()
        a.transforms, a.transformed_spots, a.cull_setup = transforms.ptr, spots.ptr, setup.ptr
        a.params, a.spot_push, a.setup_push, a.z_push = C.addressof(prm), C.addressof(push), C.addressof(ps), C.addressof(pz)
        a.light_ranges, a.range_out = zr_buf.ptr, ranges.ptr
        host = []
        if pinned:
            for field, arr in (("src_lights", self.lights.view(np.uint8)[:self.n * 48]), ("src_models", self.model.view(np.uint8).reshape(-1)[:self.n * 48]),
                               ("src_type_mask", self.type_mask.view(np.uint8)[:4 * ((self.n + 31) // 32)]), ("src_ranges", zr.view(np.uint8).reshape(-1))):
                ptr = C.c_void_p()
                gr.check(lib.gr_alloc_host(h, max(arr.size, 16), C.byref(ptr)))
                C.memmove(ptr.value, np.ascontiguousarray(arr).ctypes.data, arr.size)
                setattr(a, field, ptr.value)
                host.append(ptr)
        else:
            zr_buf.upload(zr)
        gr.check(lib.gr_cluster_front(h, None, a))
        gr.check(lib.gr_cluster_binning(h, None, transforms.ptr, setup.ptr, bitmask.ptr, prm))
        gr.sync()
        for ptr in host:
            gr.check(lib.gr_free_host(h, ptr))
        return {"transforms": transforms, "spots": spots, "setup": setup, "bitmask": bitmask, "range": ranges, "zr": zr, "zr_buf": zr_buf}

    def lighting_args(self, gr, dev, flags, alias_emissive=True):
        w, h = self.w, self.h
        imgs = {
            "albedo": capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8B8A8_SRGB).upload(self.gbuf["albedo"]),
            "normal": capi.DeviceImage(gr, w, h, capi.FORMAT_A2B10G10R10_UNORM_PACK32).upload(self.gbuf["normal"]),
            "pbr": capi.DeviceImage(gr, w, h, capi.FORMAT_R8G8_UNORM).upload(self.gbuf["pbr"]),
            "depth": capi.DeviceImage(gr, w, h, capi.FORMAT_D32_SFLOAT).upload(self.gbuf["depth"]),
            "hdr": capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16B16A16_SFLOAT).upload(self.gbuf["emissive"]),
        }
        a = capi.LightingArgs()
        a.albedo, a.normal, a.pbr, a.depth, a.hdr = (imgs[k].desc for k in ("albedo", "normal", "pbr", "depth", "hdr"))
        if alias_emissive:
            a.emissive = imgs["hdr"].desc  # reference semantics: blend read-modify-write on one attachment
        else:
            imgs["emissive"] = capi.DeviceImage(gr, w, h, capi.FORMAT_R16G16B16A16_SFLOAT).upload(self.gbuf["emissive"])
            imgs["hdr"].upload(np.full_like(self.gbuf["emissive"], 0x7e00))  # NaN-fill: every pixel must be written
            a.emissive = imgs["emissive"].desc
        a.inv_view_projection[:] = self.rp[80:96]
        col2 = self.rp[80 + 8:80 + 12]
        a.directional.inv_view_proj_col2[:] = col2
        a.directional.color[:] = synth.DIRECTIONAL_COLOR
        a.directional.direction[:] = synth.DIRECTIONAL_DIRECTION
        a.directional.camera_pos[:] = self.rp[96:99]
        a.directional.camera_front[:] = self.rp[99:102]
        a.directional.inv_resolution[:] = (1.0 / w, 1.0 / h)
        a.clustering.inv_view_proj_col2[:] = col2
        a.clustering.camera_pos[:] = self.rp[96:99]
        a.clustering.inv_resolution[:] = (1.0 / w, 1.0 / h)
        a.cluster = self.cluster_params_struct()
        a.transforms = dev["transforms"].ptr
        a.bitmask = dev["bitmask"].ptr
        a.range = dev["range"].ptr
        a.flags = flags
        return a, imgs
