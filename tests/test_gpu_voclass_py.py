"""The Python VOClass mirror driving the reference's main loop (Source/main.cpp:36-60) on the GPU: same trajectory as the
oracle's loop, same in-place side effects as the reference's methods, same point-cloud file as the host writer."""
import numpy as np
import pytest

from visualodometry_amd import kitti, synth
from visualodometry_amd.voclass import VOClass
from oracle import binding as ob

pytestmark = pytest.mark.gpu

W, H = 320, 96


def test_main_loop_with_the_python_mirror(tmp_path):
    seq = synth.Sequence(seed=1234, width=W, height=H)
    frames = [seq.frame(i) for i in range(4)]
    VO = VOClass(W, H, num_disparities=64, max_features=4096)
    VO.setProjectionMatrices(seq.P0, seq.P1)
    estimatedTrajectory = [np.zeros(3)]
    VO.imgLT2, VO.imgRT2 = frames[0]
    for i in range(len(frames) - 1):
        VO.imgLT1, VO.imgRT1 = VO.imgLT2, VO.imgRT2
        VO.imgLT2, VO.imgRT2 = frames[i + 1]
        disparityMapT1 = VO.computeDisparity(VO.imgLT1, VO.imgRT1)
        before = disparityMapT1.copy()
        depthMapT1 = VO.computeDepthMap(disparityMapT1)
        # the reference patches its argument in place (0 and -1 become 0.1f)
        bad = (before == 0.0) | (before == -1.0)
        assert bad.any() and np.all(disparityMapT1[bad] == np.float32(0.1)) and np.array_equal(disparityMapT1[~bad], before[~bad])
        if i == 0:
            n = VO.writeToPLY(str(tmp_path / "gpu.ply"))
            written, _ = kitti.write_depth_ply(str(tmp_path / "host.ply"), depthMapT1, VO.imgLT1)
            assert n == written
            assert (tmp_path / "gpu.ply").read_text().splitlines()[10:] == (tmp_path / "host.ply").read_text().splitlines()[10:]
        featurePointsT1 = VO.getFeaturesFAST(VO.imgLT1)
        n_fast = len(featurePointsT1)
        featurePointsT1, featurePointsT2 = VO.matchFeatureKLT(featurePointsT1)
        assert 0 < len(featurePointsT1) == len(featurePointsT2) <= n_fast
        t = VO.estimateMotion(featurePointsT1, featurePointsT2, depthMapT1)
        assert t.shape == (3, 1) and np.array_equal(t[:, 0], VO.poseRt[:3, 3])
        estimatedTrajectory.append(t[:, 0].copy())
    VO.close()
    op = ob.default_params(W, H, num_disparities=64)
    pose = np.eye(4)
    want = [np.zeros(3)]
    for i in range(len(frames) - 1):
        ob.step(frames[i][0], frames[i][1], frames[i + 1][0], frames[i + 1][1], op, seq.P0, seq.P1, pose)
        want.append(pose[:3, 3].copy())
    assert np.abs(np.array(estimatedTrajectory) - np.array(want)).max() <= 1e-3


def test_depth_before_disparity_is_refused():
    VO = VOClass(W, H, num_disparities=64, max_features=4096)
    with pytest.raises(Exception):
        VO.computeDepthMap(np.zeros((H, W), dtype=np.float32))
    VO.close()
