"""Import the REFERENCE's unmodified Python (options, model shell) on top of the ``models`` overlay of pointnerf_amd, in this
authoring container (where /root/reference exists).  Third-party packages the reference imports at module level but that
are absent here (image IO, MVSNet ops, plotting ...) are replaced by inert stub modules: none of them is on the hot path."""
import importlib
import importlib.abc
import importlib.machinery
import os
import shlex
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("POINTNERF_REFERENCE", "/root/reference")
STUBBED = ("torchvision", "imageio", "cv2", "PIL", "matplotlib", "skimage", "lpips", "kornia", "inplace_abn", "torch_scatter",
           "pycuda", "h5py", "plyfile", "open3d", "tqdm_never", "scipy_never", "tensorboardX", "dominate", "visdom", "pytorch3d",
           "mpl_toolkits", "trimesh", "pyhocon", "seaborn", "warmup_scheduler", "torch_optimizer", "pytorch_lightning", "test_tube")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _Stub(self.__name__ + "." + name)
        v.__path__ = []
        setattr(self, name, v)
        return v

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + "()")

    def __mro_entries__(self, bases):       # `class X(stub.Base)` in reference code
        return (object,)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in STUBBED:
            try:
                if name.split(".")[0] not in sys.modules or not isinstance(sys.modules[name.split(".")[0]], _Stub):
                    if importlib.machinery.PathFinder.find_spec(name.split(".")[0]) is not None:
                        return None         # really installed: use it
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def install():
    """overlay first on sys.path, the reference behind it, stubs for the absent third-party packages"""
    ov = os.path.join(ROOT, "pointnerf_amd", "overlay")
    for p in (ov, ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, ov)
    sys.path.insert(1, ROOT)
    os.environ["POINTNERF_REFERENCE"] = REF
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())


def script_argv(script="dev_scripts/w_n360/lego_cuda.sh"):
    """the command line the reference's own launch script builds (its shell variables expanded by bash, the python call echoed)"""
    text = open(os.path.join(REF, script)).read()
    text = text.replace("cd run", "").replace("python train_ft_nonstop.py", "echo PNERF_ARGS").replace("python3 train_ft.py", "echo PNERF_ARGS")
    out = subprocess.run(["bash", "-c", text], capture_output=True, text=True).stdout
    line = [ln for ln in out.splitlines() if ln.startswith("PNERF_ARGS")][-1]
    return shlex.split(line)[1:]


def parse_options(extra=()):
    """opt exactly as run/train_ft.py gets it: TrainOptions().parse() of the reference on the script's command line"""
    install()
    argv = script_argv() + list(extra)
    old = sys.argv
    sys.argv = ["train_ft.py"] + argv
    try:
        importlib.import_module("models")
        opts = importlib.import_module("options")
        return opts.TrainOptions().parse()
    finally:
        sys.argv = old
