"""Python mirror of the reference's `VOClass` (Include/VOClass.h:62-123) on top of the C ABI.

Same method names, argument meaning and side effects as the reference's hot-path methods, on numpy arrays
instead of cv::Mat / std::vector<cv::Point2f>:

    vo = VOClass(width, height); vo.setProjectionMatrices(P0, P1)
    vo.imgLT1, vo.imgRT1, vo.imgLT2, vo.imgRT2 = ...            (public members, Include/VOClass.h:65-66)
    disp  = vo.computeDisparity(vo.imgLT1, vo.imgRT1)            (Source/VOClass.cpp:231)
    depth = vo.computeDepthMap(disp)                             (Source/VOClass.cpp:312; patches disp in place)
    f1    = vo.getFeaturesFAST(vo.imgLT1)                        (Source/VOClassFeatures.cpp:7)
    f1, f2 = vo.matchFeatureKLT(f1)                              (Source/VOClassFeatures.cpp:54; f1 is replaced)
    t     = vo.estimateMotion(f1, f2, depth)                     (Source/VOClass.cpp:370; updates vo.poseRt)

The C++ form of the same class is include/VOClass.h.
"""
import numpy as np
from . import abi


class VOClass:
    def __init__(self, width=1241, height=376, device=0, **params):
        self.frameW, self.frameH = int(width), int(height)
        self.params = abi.default_params(width, height, **params)
        self.ctx = abi.Context(self.params, device)
        self.imgLT1 = self.imgRT1 = self.imgLT2 = self.imgRT2 = None
        self.poseRt = np.eye(4, dtype=np.float64)
        self.projectionCL = np.zeros((3, 4))
        self.projectionCR = np.zeros((3, 4))

    # state that getProjectionMatrices (Source/VOClass.cpp:102) fills from calib.txt
    def setProjectionMatrices(self, P0, P1):
        self.projectionCL = np.array(P0, dtype=np.float64).reshape(3, 4)
        self.projectionCR = np.array(P1, dtype=np.float64).reshape(3, 4)
        self.ctx.set_calib(self.projectionCL, self.projectionCR)

    def _sync_images(self, lt1=None, rt1=None):
        """Uploads the images a method is about to work on.  Always: the reference re-reads its cv::Mat arguments and
        members on every call, so a caller that refills the same arrays in place must see the new pixels used."""
        lt1 = self.imgLT1 if lt1 is None else lt1
        rt1 = self.imgRT1 if rt1 is None else rt1
        lt2 = self.imgLT2 if self.imgLT2 is not None else lt1
        rt2 = self.imgRT2 if self.imgRT2 is not None else rt1
        self.ctx.set_images(lt1, rt1, lt2, rt2)

    def computeDisparity(self, leftImg, rightImg):
        self._sync_images(leftImg, rightImg)
        self.ctx.compute_disparity()
        return self.ctx.disparity32(0)

    def computeDepthMap(self, disparityMap):
        # the reference converts the map it is handed (Source/VOClass.cpp:312-366), whatever produced it
        self.ctx.set_disparity32(disparityMap, 0)
        self.ctx.compute_depth_map()
        # ... and patches the caller's map in place (shared cv::Mat header)
        disparityMap[(disparityMap == 0.0) | (disparityMap == -1.0)] = np.float32(0.1)
        return self.ctx.depth(0)

    def writeToPLY(self, depthMap, colors, depthThresh=None, numVertices=None, plyFilePath="depth.ply"):
        """VOClass::writeToPLY(depthMap, colors, depthThresh, numVertices) (Source/VOClassUtils.cpp:111-157): "row col -z g g g"
        for every pixel of depthMap that is not above depthThresh, colours from the image handed over.  Works on its
        arguments like the reference (callable at any point of the loop); the vertices are compacted on the GPU in the
        writer's order.  numVertices is the header's count (default: the vertices written); depthThresh must be the
        context's depth_thresh (3000 in the reference's only call, Source/VOClass.cpp:358)."""
        from . import kitti
        if depthThresh is not None and int(depthThresh) != int(self.params.depth_thresh):
            raise ValueError("depthThresh %s differs from the context's depth_thresh %d" % (depthThresh, self.params.depth_thresh))
        self._sync_images(colors, None)
        self.ctx.set_depth(depthMap, 0)
        return kitti.write_cloud_ply(plyFilePath, self.ctx.depth_cloud(0), numVertices)

    def getFeaturesFAST(self, img):
        self._sync_images(img, None)
        self.ctx.detect_fast()
        return self.ctx.features(0)

    def matchFeatureKLT(self, featurePointsLT1):
        """Returns (featurePointsLT1 after the in/out overwrite, featurePointsLT2)."""
        self._sync_images()
        self.ctx.set_features(featurePointsLT1, 0)
        self.ctx.match_klt()
        return self.ctx.matches(0)

    def estimateMotion(self, featurePointsT1, featurePointsT2, depthMap):
        """depthMap is required, as in the reference (Include/VOClass.h:98-100): depths are looked up in the map handed over
        (Source/VOClass.cpp:418), whatever the device computed last."""
        assert len(featurePointsT1) == len(featurePointsT2)
        self.ctx.set_matches(featurePointsT1, featurePointsT2, 0)
        self.ctx.set_depth(depthMap, 0)
        self.ctx.triangulate()
        Rt, info = self.ctx.estimate_motion()
        self.lastInfo = info[0]
        self.poseRt = abi.integrate_pose(self.poseRt, Rt[0])
        return self.poseRt[:3, 3].copy().reshape(3, 1)

    def close(self):
        self.ctx.close()
