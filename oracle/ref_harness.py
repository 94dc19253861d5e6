"""TEST INFRASTRUCTURE ONLY -- imports the *reference's own* Python modules on CPU.

Works only in the build container (needs /root/reference); nothing under tests/ marked
gpu, bench.py or smoke() may import this file.  It exists so that
oracle/gen_golden.py can run the reference verbatim and freeze its outputs into
tests/golden/*.npz (pre-seed sys.modules with permissive stubs for the
third-party / CUDA-extension imports, then import model.*, utils, MCAcc).
"""
import sys
import types
import importlib
import warnings

REF_ROOT = "/root/reference"


class _Anything:
    """Permissive stand-in: any attribute / call yields another stand-in."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__path__ = []  # behave like a package so that sub-imports resolve
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a namespace with the reference modules imported unmodified (CPU)."""
    import torch  # noqa: F401

    if "selfrecon_reference_loaded" in sys.modules:
        return sys.modules["selfrecon_reference_loaded"]
    warnings.filterwarnings("ignore")

    class CamerasBase(torch.nn.Module):  # pytorch3d.renderer.cameras.CamerasBase must be a real class
        def __init__(self, *a, **k):
            """pytorch3d's TensorProperties keeps the constructor's keyword tensors as attributes (focal_length, principal_point, R, T,
            image_size ...): that much is needed for RectifiedPerspectiveCameras(...).view_rays / cam_pos to run."""
            super().__init__()
            for name, value in k.items():
                if name == "image_size" and not torch.is_tensor(value):
                    value = torch.tensor(value)
                object.__setattr__(self, name, value)

        def to(self, device):
            return self

    for name in [
        "pytorch3d", "pytorch3d.structures", "pytorch3d.loss", "pytorch3d.io", "pytorch3d.renderer",
        "pytorch3d.renderer.mesh", "pytorch3d.renderer.mesh.renderer", "pytorch3d.transforms",
        "pytorch3d.renderer.points", "pytorch3d.renderer.points.rasterizer", "pytorch3d.renderer.utils",
        "torch_scatter", "cv2", "trimesh", "openmesh", "MCGpu", "FastMinv", "GridSamplerMine",
        "interp2x_boundary3d", "pyhocon", "h5py",
    ]:
        _stub(name)
    _stub("pytorch3d.renderer.cameras", CamerasBase=CamerasBase)
    sys.modules["pytorch3d.renderer"].cameras = sys.modules["pytorch3d.renderer.cameras"]

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.ModuleType("selfrecon_reference_loaded")
    ns.network = importlib.import_module("model.network")
    ns.Deformer = importlib.import_module("model.Deformer")
    ns.RenderNet = importlib.import_module("model.RenderNet")
    ns.Embedder = importlib.import_module("model.Embedder")
    ns.utils = importlib.import_module("utils")
    ns.rutils = importlib.import_module("utils.utils")
    ns.FindSurfacePs = importlib.import_module("utils.FindSurfacePs")
    ns.MCAcc = importlib.import_module("MCAcc")
    ns.seg3d = importlib.import_module("MCAcc.seg3d_lossless")
    ns.smpl_util = importlib.import_module("smpl_pytorch.util")
    sys.modules["selfrecon_reference_loaded"] = ns
    return ns


if __name__ == "__main__":
    ref = load_reference()
    net = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
    import torch
    x = torch.randn(5, 3)
    print(net(x, 1.0).shape, net.rendcond.shape)
