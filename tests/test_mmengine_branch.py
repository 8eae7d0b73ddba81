"""CPU: the `USING_MMENGINE` branch of probpose_code_amd/registry.py (VERDICT r1 missing item 5). Neither mmengine nor
mmpose is installed here, so a stand-in ``mmpose.registry`` module - registries with mmengine's ``register_module(name,
force, module)`` / ``build(cfg)`` / ``get`` contract (mmengine/registry/registry.py [3P]), pre-populated with
"reference" classes under the reference's names as ``mmpose/registry.py:50,92`` + the ``@MODELS.register_module()``
decorators do - is injected into ``sys.modules`` of a fresh interpreter before ``import probpose_code_amd``, which is
what ``custom_imports=dict(imports=["probpose_code_amd"])`` triggers in a real MMPose."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = textwrap.dedent('''
    import sys, types
    class FakeMMEngineRegistry:                      # mmengine.registry.Registry, the part a drop-in touches
        def __init__(self, name): self.name, self._d = name, {}
        @property
        def module_dict(self): return self._d
        def get(self, key): return self._d.get(key.split(".", 1)[1] if "." in key and key not in self._d else key)
        def register_module(self, name=None, force=False, module=None):
            def reg(cls):
                for n in ([cls.__name__] if name is None else ([name] if isinstance(name, str) else list(name))):
                    if n in self._d and not force:
                        raise KeyError(f"{n} is already registered in {self.name}")
                    self._d[n] = cls
                return cls
            return reg if module is None else reg(module)
        def build(self, cfg, *a, **kw):
            cfg = dict(cfg); cls = self.get(cfg.pop("type")); assert cls is not None; return cls(*a, **cfg)
    reg = types.ModuleType("mmpose.registry")
    reg.MODELS, reg.KEYPOINT_CODECS = FakeMMEngineRegistry("model"), FakeMMEngineRegistry("keypoint codec")
    class RefEstimator: pass
    class RefHead: pass
    class RefCodec: pass
    class RefPre: pass
    reg.MODELS.register_module(name="TopdownPoseEstimator", module=RefEstimator)
    reg.MODELS.register_module(name="ProbMapHead", module=RefHead)
    reg.MODELS.register_module(name="PoseDataPreprocessor", module=RefPre)
    reg.KEYPOINT_CODECS.register_module(name="ProbMap", module=RefCodec)
    pkg = types.ModuleType("mmpose"); pkg.__path__ = []; pkg.registry = reg
    sys.modules["mmpose"], sys.modules["mmpose.registry"] = pkg, reg
    sys.path.insert(0, %r)
''') % ROOT


def _run(body, **env):
    e = {k: v for k, v in os.environ.items() if k != "PROBPOSE_MI355X_OVERRIDE"}
    e.update(env)
    r = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_registers_into_the_real_registries_beside_the_reference(lib_built):
    out = _run('''
        import probpose_code_amd as pp
        from probpose_code_amd import registry as R
        assert R.USING_MMENGINE and R.MODELS is reg.MODELS and R.KEYPOINT_CODECS is reg.KEYPOINT_CODECS
        for n in ("TopdownPoseEstimatorMI355X", "ProbMapHeadMI355X", "VisionTransformerMI355X", "PoseDataPreprocessorMI355X"):
            assert n in reg.MODELS.module_dict, n
        assert "ProbMapMI355X" in reg.KEYPOINT_CODECS.module_dict
        # without the override switch the reference's own names still point at the reference classes
        assert reg.MODELS.get("TopdownPoseEstimator") is RefEstimator and reg.MODELS.get("ProbMapHead") is RefHead
        assert reg.KEYPOINT_CODECS.get("ProbMap") is RefCodec and reg.MODELS.get("PoseDataPreprocessor") is RefPre
        # a derived config (INTEGRATION.md 1) builds the whole estimator THROUGH the real registries
        cfg = pp.Config.fromfile("%s/configs/td-pm_ProbPose-small_mi355x_coco-256x192.py")
        m = dict(cfg.model)
        m["type"] = "TopdownPoseEstimatorMI355X"
        m["data_preprocessor"] = dict(m["data_preprocessor"], type="PoseDataPreprocessorMI355X")
        m["backbone"] = dict(m["backbone"], type="VisionTransformerMI355X")
        m["head"] = dict(m["head"], type="ProbMapHeadMI355X", decoder=dict(m["head"]["decoder"], type="ProbMapMI355X"))
        model = reg.MODELS.build(m)
        assert type(model).__name__ == "TopdownPoseEstimator" and type(model.head.decoder).__name__ == "ProbMap"
        assert model.head.decoder.support_batch_decoding
        from probpose_code_amd import synthetic
        sd = synthetic.synthetic_state_dict("small", seed=0)
        res = model.load_state_dict(sd, strict=True)
        print("OK", len(model.state_dict()))
    ''' % ROOT)
    assert out.strip().startswith("OK")


def test_override_switch_replaces_the_reference_names(lib_built):
    out = _run('''
        import probpose_code_amd as pp
        from probpose_code_amd import pose_estimators as PE, codecs as C
        assert reg.MODELS.get("TopdownPoseEstimator") is PE.TopdownPoseEstimator
        assert reg.MODELS.get("ProbMapHead") is PE.ProbMapHead and reg.MODELS.get("PoseDataPreprocessor") is PE.PoseDataPreprocessor
        assert reg.KEYPOINT_CODECS.get("ProbMap") is C.ProbMap
        assert reg.MODELS.get("mmpretrain.VisionTransformer") is PE.VisionTransformer  # the config's cross-scope name (:57)
        # the REFERENCE config's model block, unedited, now lands on the MI355X classes
        cfg = pp.Config.fromfile("%s/configs/td-pm_ProbPose-small_mi355x_coco-256x192.py")
        model = reg.MODELS.build(dict(cfg.model))
        assert isinstance(model, PE.TopdownPoseEstimator)
        print("OK")
    ''' % ROOT, PROBPOSE_MI355X_OVERRIDE="1")
    assert out.strip() == "OK"
