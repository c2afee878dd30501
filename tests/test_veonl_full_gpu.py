"""BASELINE configs[3] / [4] at their REAL size in the GPU suite:
VEON-L = SAN on CLIP ViT-L/14-336 + DepthAnythingV2 ViT-L, 6 cameras 256x704, D = 88,
C = 256, 200x200x16 voxels.

* the whole forward replayed from ONE hipGraph == the eager two-stream forward, in
  both 16-bit flavours (configs[4] names fp16, configs[2] bf16);
* configs[3]'s camera sharding on a single rank: ``forward_camera_sharded`` (all six
  cameras, no group) == ``forward``; ``CameraShardedStep`` (hipGraph segments around
  the collectives) through a ONE-rank RCCL group, all-reduce and reduce-scatter +
  sharded max-pool + all-gather, == ``forward``; disjoint camera halves add up.

Tolerances are those of tests/test_graphs_gpu.py: 2e-2 of the logit scale (MIOpen
may pick other algorithms under capture; the un-pooled fp32 volume is max-pooled and
packed to 16 bit at another point than in the fused path), class-map agreement >= 99 %."""
import socket

import pytest
import torch

from veon_amd import half, synthetic
from veon_amd.graphs import GraphedCallable

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SIZE = (256, 704)


def _net_and_inputs():
    from veon_amd.models.veon_occ import VeonOccupancyPath
    torch.manual_seed(0)
    net = VeonOccupancyPath(input_size=SIZE, **VeonOccupancyPath.VEON_L).to(DEV).eval()
    geom = [t.to(DEV) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, SIZE))]
    images = torch.randn(1, 6, 3, *SIZE, device=DEV)
    return net, images, geom


def _close(got, ref, what):
    for k in ('sem_occ', 'bin_occ'):
        err = (got[k].float() - ref[k].float()).abs().max().item()
        assert err <= 2e-2 * max(1.0, ref[k].abs().max().item()), (what, k, err)
    agree = (got['sem_occ'].argmax(1) == ref['sem_occ'].argmax(1)).float().mean().item()
    assert agree >= 0.99, (what, agree)


@pytest.mark.parametrize('flavour', ['bf16', 'fp16'])
def test_veon_l_full_size_graph_matches_eager(flavour):
    with half.use(flavour):
        net, images, geom = _net_and_inputs()
        assert len(net.clip_trunk.resblocks) == 24 and net.depth_model.pretrained.embed_dim == 1024
        with torch.no_grad():
            want = {k: v.clone() for k, v in net(images, geom).items()}
            assert want['sem_occ'].shape == (1, 17, 16, 200, 200)
            assert all(torch.isfinite(v.float()).all() for v in want.values())
            graphed = GraphedCallable(lambda im: net(im, geom), (images,))
            images2 = torch.randn_like(images)
            want2 = {k: v.clone() for k, v in net(images2, geom).items()}
            for im, ref in ((images, want), (images2, want2), (images, want)):
                _close(graphed(im), ref, 'graph replay ' + flavour)
            assert (want2['sem_occ'] - want['sem_occ']).abs().max().item() > 0


@pytest.fixture
def one_rank_rccl_group():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0,
                            world_size=1, device_id=torch.device(DEV))
    try:
        yield dist.group.WORLD
    finally:
        dist.destroy_process_group()


def test_veon_l_full_size_camera_sharding_on_one_rank(one_rank_rccl_group):
    from veon_amd.models.veon_occ import CameraShardedStep
    net, images, geom = _net_and_inputs()
    with torch.no_grad():
        want = {k: v.clone() for k, v in net(images, geom).items()}
        _close(net.forward_camera_sharded(images, geom), want, 'eager sharded, all-reduce')
        _close(net.forward_camera_sharded(images, geom, reduce_dtype=torch.bfloat16),
               want, 'eager sharded, bf16 message')
        # V = sum over cameras: two half rigs add up to the whole
        v = net.lift_cameras(images, geom, 0, 3) + net.lift_cameras(images, geom, 3, 6)
        _close(net.from_volume(v), want, 'sum of camera halves')
        # the exchange of reduce='scatter' through the 1-rank RCCL group
        pooled = net._scatter_pool_gather(v, 1, one_rank_rccl_group, torch.bfloat16)
        assert pooled.shape == (1, 256, 8, 100, 100)
        assert torch.equal(pooled, net._max_pool(v).to(torch.bfloat16).float())
        del v, pooled
        images2 = torch.randn_like(images)
        want2 = {k: v_.clone() for k, v_ in net(images2, geom).items()}
        for reduce in ('allreduce', 'scatter'):
            step = CameraShardedStep(net, images, geom, group=one_rank_rccl_group,
                                     reduce_dtype=torch.bfloat16, reduce=reduce)
            assert step.cameras == (0, 6) and step.scatter == (reduce == 'scatter')
            for im, ref in ((images, want), (images2, want2), (images, want)):
                _close(step(im), ref, 'graph segments, ' + reduce)
            del step
