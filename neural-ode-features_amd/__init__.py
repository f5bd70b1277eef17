"""neural-ode-features_amd -- MI355X-native drop-in for the ODE-block hot path of
fabiocarrara/neural-ode-features.

Aliasable as `torchdiffeq` (it exports `odeint` and `odeint_adjoint`), so the
reference's `model.py` runs unchanged on top of the HIP library:

    import neural_ode_features_amd, sys
    sys.modules['torchdiffeq'] = neural_ode_features_amd      # before `import model`
"""
from .integrate import odeint, odeint_adjoint, odefunc_forward, odefunc_vjp  # noqa: F401
from .modules import ConcatConv2d, ODEBlock, ODEfunc, normalization  # noqa: F401
from .odenet import FCClassifier, ODEDownsample, ODEDownsample2, ODENet, ResBlock, StackedODENet  # noqa: F401
from . import dp, graphs, optim  # noqa: F401
from .optim import FusedSGD  # noqa: F401
from .head import cross_entropy, linear, linear_cross_entropy  # noqa: F401

__all__ = ['odeint', 'odeint_adjoint', 'ODEBlock', 'ODEfunc', 'ConcatConv2d', 'ODENet', 'StackedODENet',
           'ODEDownsample', 'ODEDownsample2', 'dp']
