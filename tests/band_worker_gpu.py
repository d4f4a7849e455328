"""One rank of the multi-PROCESS row-band test on the GPU box (launched by test_gpu_multiprocess.py through torch.distributed.run,
every rank on GPU 0, collectives through tests/rccl_shim): the set-up bench.py makes -- gloo as the control plane, two communicators,
the 1/8 level gathered in the frame, the output bands beside it -- a few un-synchronised frames, then every rank's assembled
backbuffer, 1/8 level and exposure go to disk for the comparison with the single-process frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from granite_amd import app as gapp, synth
    w, h, lights, frames, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    post_aa = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    pre_aa = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    reach = int(sys.argv[8]) if len(sys.argv) > 8 else 0  # gra_config.taa_history_reach_rows
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    cam = synth.Camera(w, h)
    a = gapp.Application(w, h, device=0, strip_index=rank, strip_count=world, post_aa=post_aa, pre_aa=pre_aa, taa_history_reach_rows=reach)
    assert not reach or a.strip_plan()["taa_exchange_rows"] > 0
    if pre_aa:  # the temporal resolve jitters the projection: it needs the camera, and motion vectors
        a.set_camera(np.ascontiguousarray(cam.P.T, np.float32).reshape(16), np.ascontiguousarray(cam.V.T, np.float32).reshape(16))
    else:
        a.set_render_parameters(cam.render_params())
    a.set_lights(synth.make_lights(cam, lights))
    a.upload_gbuffer(synth.make_gbuffer(cam), synth.make_motion_vectors(w, h) if pre_aa else None)
    ids = [gapp.Application.comm_create_unique_id() if rank == 0 else None, gapp.Application.comm_create_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    a.comm_init(ids[0], rank, world)
    a.comm_init_output(ids[1], rank, world)
    a.render_frames(frames, sync=False)   # pipelined: the output gathers of several frames are in flight beside the frames
    a.sync()
    np.savez(out.format(rank=rank), backbuffer=a.read_backbuffer(), d1=a.read("downsample-1"), lum=a.read("average-luminance"))
    dist.barrier()
    a.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
