"""probpose_code_amd -- MI355X-native (gfx950 / CDNA4) implementation of the ProbPose
top-down inference hot path behind the MMPose registry/config surface.

Importing the package registers the MI355X components (see ``registry.py``); this is what
``custom_imports = dict(imports=["probpose_code_amd"])`` in an MMPose config triggers.
The compute lives in ``libprobpose_mi355x.so`` (hand-written HIP, C ABI declared in
``include/probpose_mi355x.h``); the Python here is the host-side mirror of the reference's
operator interface for this path and nothing else.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .codecs import BaseKeypointCodec, ProbMap, oks_kernel_taps  # noqa: F401
from .config import Config  # noqa: F401
from .engine import ProbPoseEngine, domain_report  # noqa: F401
from .pipeline import StepPipeline  # noqa: F401
from .pose_estimators import (  # noqa: F401
    PoseDataPreprocessor,
    ProbMapHead,
    TopdownPoseEstimator,
    VisionTransformer,
    build_pose_estimator,
)
from .registry import KEYPOINT_CODECS, MODELS, TRANSFORMS  # noqa: F401
from .structures import InstanceData, PixelData, PoseDataSample  # noqa: F401

__version__ = "0.1.0"
