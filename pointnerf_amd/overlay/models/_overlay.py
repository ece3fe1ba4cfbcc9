"""Plumbing of the ``models`` overlay: where the reference checkout is, and how a module of the overlay pulls the
reference's own module of the same path in (for the names the overlay does not replace)."""
import importlib.util
import os
import sys

REF_ROOT = os.environ.get("POINTNERF_REFERENCE", "")        # no default: the overlay never guesses where a checkout might be
REF_MODELS = os.path.join(REF_ROOT, "models")


def require_reference():
    if not REF_ROOT or not os.path.isdir(REF_MODELS):
        raise ImportError("pointnerf_amd overlay: the Point-NeRF checkout was not found at %r; set POINTNERF_REFERENCE to it "
                          "(the overlay replaces only the hot-path modules, everything else is the reference's own code)" % REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.append(REF_ROOT)          # `utils`, `data`, `run` of the reference, behind the overlay


def extend_path(pkg_path, *rel):
    """Let package lookups fall through to the reference's directory of the same name."""
    require_reference()
    d = os.path.join(REF_MODELS, *rel)
    if d not in pkg_path:
        pkg_path.append(d)


def load_reference_module(rel_path, alias):
    """Import the reference's file models/<rel_path> under the module name ``alias`` (its relative imports resolve inside
    the ``models`` package, i.e. through the overlay)."""
    require_reference()
    if alias in sys.modules:
        return sys.modules[alias]
    spec = importlib.util.spec_from_file_location(alias, os.path.join(REF_MODELS, rel_path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod
