"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls without a GPU)."""
import ctypes
import glob
import os
import re

from granite_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for header in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(header).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(gr[a-z]*_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_headers_declare_entry_points():
    names = declared_functions()
    assert "gr_create" in names and "gr_lighting" in names and "gr_tonemap" in names


def test_library_exports_every_declared_symbol():
    for path in glob.glob(os.path.join(ROOT, "granite_amd", "lib", "*.so")):
        ctypes.CDLL(path)  # must dlopen without a GPU
    libs = [ctypes.CDLL(p) for p in glob.glob(os.path.join(ROOT, "granite_amd", "lib", "*.so"))]
    assert libs, "no built libraries under granite_amd/lib — run __graft_entry__.build()"
    missing = []
    for name in declared_functions():
        if not any(hasattr(lib, name) for lib in libs):
            missing.append(name)
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_ctypes_binding_matches_library():
    lib = capi.load_library()
    assert lib.gr_abi_version() == 1
    for name in capi.EXPORTED_SYMBOLS:
        assert hasattr(lib, name)


def test_struct_layouts_match_reference_contracts():
    # SURVEY.md Appendix A
    assert ctypes.sizeof(capi.ClusterParams) == 176
    assert capi.ClusterParams.clip_scale.offset == 64
    assert capi.ClusterParams.camera_base.offset == 80
    assert capi.ClusterParams.camera_front.offset == 96
    assert capi.ClusterParams.xy_scale.offset == 112
    assert capi.ClusterParams.resolution_xy.offset == 120
    assert capi.ClusterParams.inv_resolution_xy.offset == 128
    assert capi.ClusterParams.num_lights.offset == 136
    assert capi.ClusterParams.z_max_index.offset == 156
    assert capi.ClusterParams.z_scale.offset == 160
    assert ctypes.sizeof(capi.PushBloomThreshold) == 16
    assert ctypes.sizeof(capi.PushBloomDownsample) == 28
    assert ctypes.sizeof(capi.PushBloomUpsample) == 24
    assert ctypes.sizeof(capi.PushLuminance) == 20
    assert ctypes.sizeof(capi.PushTonemap) == 4
    assert ctypes.sizeof(capi.PushSpotTransform) == 100
    assert ctypes.sizeof(capi.PushClusterSetup) == 68
    assert ctypes.sizeof(capi.PushZRange) == 12
    assert capi.PushDirectional.inv_resolution.offset == 80
    assert capi.PushDirectional.camera_front.offset == 64
    assert capi.PushDirectional.direction.offset == 48
    assert capi.PushClustering.inv_resolution.offset == 32
    assert capi.TRANSFORMS_OFFSET_MODEL == 4096 * 48 + 4096 * 64
    assert capi.TRANSFORMS_OFFSET_TYPE_MASK == capi.TRANSFORMS_OFFSET_MODEL + 4096 * 48
    assert capi.TRANSFORMS_SIZE == capi.TRANSFORMS_OFFSET_DECALS + 4096 * 48


def test_headers_assert_their_own_layouts_in_c_and_cxx(tmp_path):
    """include/granite_hip.h carries static assertions on the size and member offsets of every push-constant block and of the cluster
    UBO: a C11 and a C++17 translation unit that only includes the headers must compile."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "granite_hip.h"\n#include "granite_app.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "abi_c.o")])
    subprocess.check_call(["g++", "-std=c++17", "-x", "c++", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "abi_cxx.o")])
    text = open(os.path.join(inc, "granite_hip.h")).read()
    for struct in re.findall(r"typedef struct (gr_push_\w+|gr_cluster_params|gr_light_info|gr_luminance_data)", text):
        assert f"GR_ASSERT_SIZE({struct}," in text, f"{struct} has no size assertion in the header"
