"""Hook-up with the nequip / allegro plugin machinery (SURVEY §8b "activation" and "discovery").

The reference activates an accelerated contracter through a *model modifier*: a
`@model_modifier(persistent=False) @classmethod` on `Contracter` (allegro/nn/_strided/_contract.py:253-255,
284-286) that nequip finds by name from `nequip.model.modify` in a config or `nequip-compile --modifiers ...`
(docs/guide/triton.md:18-26), and packages announce themselves through the `nequip.extension` entry-point group
(pyproject.toml:50-51, `init_always = "allegro"`: import side effects, allegro/__init__.py:3-4).

`register()` does the same for this package, when nequip and allegro are importable:
  * `Contracter.enable_HipContracter` -- the modifier, in exactly the reference's form;
  * the package is declared external to `nequip-package` archives (allegro/_extern.py does this for
    cuequivariance), since its kernels live in a shared library, not in picklable Python.
`pair_allegro` needs no new compile target: the target only fixes the tensor contract (allegro/_compile.py:10-14,
68-74), the modifier decides which contracter the compiled model calls.

Without nequip/allegro installed (this build container, the GPU box) `register()` is a no-op that reports why;
`allegro_amd.nn.enable_HipContracter(model)` is the same modifier as a plain function.
"""
from typing import Optional

_STATUS: Optional[str] = None


def register() -> str:
    """Idempotent.  Returns a one-line status ("registered ..." or the reason nothing was done)."""
    global _STATUS
    if _STATUS is not None:
        return _STATUS
    try:
        from nequip.nn import model_modifier  # EXT
    except Exception as e:  # nequip absent (or a partial shim): nothing to hook into
        _STATUS = f"nequip not importable ({type(e).__name__}): modifier available as allegro_amd.nn.enable_HipContracter"
        return _STATUS
    from .nn import HipContracter, _enable_HipContracter_classmethod

    done = []
    # HipContracter's own classmethod in the decorated form (so that a model that already holds HipContracters
    # still answers to the modifier name)
    HipContracter.enable_HipContracter = model_modifier(persistent=False)(classmethod(_enable_HipContracter_classmethod))
    try:
        from allegro.nn._strided._contract import Contracter  # the reference's class

        if not hasattr(Contracter, "enable_HipContracter"):
            Contracter.enable_HipContracter = model_modifier(persistent=False)(classmethod(_enable_HipContracter_classmethod))
        done.append("Contracter.enable_HipContracter")
    except Exception as e:
        done.append(f"allegro not importable ({type(e).__name__})")
    try:
        from nequip.scripts._package_utils import register_libraries_as_external_for_packaging

        register_libraries_as_external_for_packaging(extern_modules=["allegro_amd"])
        done.append("extern: allegro_amd")
    except Exception:
        pass
    _STATUS = "registered: " + ", ".join(done)
    return _STATUS
