"""Generates tests/golden/gtx_reference_files.json from the .gtx files the reference ships (run in the build container,
where /root/reference exists): size, sha256 and the 64-byte header of each file.  The GTX reader/writer is pinned against
these: writing the committed payloads must reproduce the reference's files bit for bit."""
import hashlib
import json
import os

REF = "/root/reference/assets/textures"
FILES = ["smaa/area.gtx", "smaa/search.gtx", "ibl_brdf_lut.gtx"]

out = {}
for name in FILES:
    raw = open(os.path.join(REF, name), "rb").read()
    out[name] = {"size": len(raw), "sha256": hashlib.sha256(raw).hexdigest(), "header_hex": raw[:64].hex(),
                 "payload_sha256": hashlib.sha256(raw[64:]).hexdigest()}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gtx_reference_files.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
