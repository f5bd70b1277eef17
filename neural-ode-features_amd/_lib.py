"""ctypes binding of libnode_hip.so (the C ABI declared in include/node_hip.h).

This is the stub a maintainer of the reference would add next to ``model.py`` to
replace ``from torchdiffeq import odeint_adjoint, odeint`` (model.py:3) -- see
INTEGRATION.md.  There is no CPU or pure-PyTorch fallback: if the library is
missing or no HIP device is present, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'csrc', 'libnode_hip.so')
# Diagnostics build (build.py --diag): the product library plus the timing ablations, in-kernel stamps and the
# measured-and-rejected kernel variants.  Only tools/ and the `-m diag` tests ask for it, by NODE_HIP_DIAG=1 in the
# environment of their own process; nothing in the package does.
LIB_DIAG_PATH = os.path.join(HERE, 'csrc', 'libnode_hip_diag.so')

NODE_ABI_VERSION = 6
METHOD_DOPRI5, METHOD_RK4 = 0, 1
METHODS = {'dopri5': METHOD_DOPRI5, 'rk4': METHOD_RK4}

ERRORS = {
    0: 'NODE_OK', -1: 'NODE_ERR_NULL', -2: 'NODE_ERR_SHAPE', -3: 'NODE_ERR_UNSUPPORTED',
    -4: 'NODE_ERR_WORKSPACE', -5: 'NODE_ERR_MAX_STEPS', -6: 'NODE_ERR_NONFINITE',
    -7: 'NODE_ERR_DT_UNDERFLOW', -8: 'NODE_ERR_HIP', -9: 'NODE_ERR_ARG',
}

EXPORTS = [
    'node_abi_version', 'node_last_error', 'node_param_count', 'node_workspace_bytes', 'node_solve_is_resident',
    'node_odefunc_fwd', 'node_odefunc_vjp', 'node_solve_fwd', 'node_solve_adjoint',
    'node_backprop_workspace_bytes', 'node_solve_backprop',
    'node_head_fwd', 'node_head_bwd', 'node_gn_relu_fwd', 'node_gn_relu_bwd',
    'node_sgd_step', 'node_profile_begin', 'node_profile_end',
    'node_conv3x3_w4_workspace_bytes', 'node_conv3x3_w4', 'node_w4_split3', 'node_w4_pair_stats',
    'node_stem_workspace_bytes', 'node_stem_fwd', 'node_stem_bwd', 'node_stem_conv_workspace_bytes', 'node_stem_conv',
    'node_head_loss_scratch_bytes', 'node_head_loss_fwd', 'node_head_loss_bwd',
    'node_flat_workspace_bytes', 'node_flat_begin', 'node_flat_stage', 'node_flat_scalar', 'node_flat_initial_step',
    'node_flat_finish_step', 'node_flat_status_read',
]


class NodeShape(C.Structure):
    _fields_ = [('n', C.c_int32), ('c', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
                ('groups', C.c_int32), ('eps', C.c_float)]


class NodeParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        'norm1_w', 'norm1_b', 'conv1_w', 'conv1_b', 'norm2_w', 'norm2_b',
        'conv2_w', 'conv2_b', 'norm3_w', 'norm3_b')]


class NodeStats(C.Structure):
    _fields_ = [('nfe', C.c_int32), ('accepted', C.c_int32), ('rejected', C.c_int32), ('status', C.c_int32),
                ('last_dt', C.c_double), ('t_final', C.c_double), ('first_dt', C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class NodeStepRecord(C.Structure):
    _fields_ = [('done', C.c_int32), ('status', C.c_int32), ('steps', C.c_int32), ('accepted', C.c_int32),
                ('rejected', C.c_int32), ('miss', C.c_int32), ('t', C.c_double), ('dt', C.c_double), ('first_dt', C.c_double),
                ('t_prev', C.c_double), ('dt_used', C.c_double)]


class NodeSolveOpts(C.Structure):
    _fields_ = [('max_num_steps', C.c_int32), ('n_forced_dt', C.c_int32), ('forced_dt', C.POINTER(C.c_double)),
                ('record_dt', C.c_int32), ('dt_log', C.POINTER(C.c_double)), ('n_dt_log', C.POINTER(C.c_int32)),
                ('blind_steps', C.c_int32), ('record', C.c_void_p), ('miss_flag', C.c_void_p), ('grad_last_only', C.c_int32),
                ('norm_reduce', C.c_void_p), ('norm_reduce_ctx', C.c_void_p), ('norm_buf', C.c_void_p), ('norm_world', C.c_int32)]


NORM_REDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


NODE_PENDING = 1


class NodeProfile(C.Structure):
    _fields_ = [('launches', C.c_int64 * 9), ('total_ms', C.c_double * 9), ('flops', C.c_double * 9)]


class NodeSgdTensor(C.Structure):
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('momentum_buf', C.c_void_p), ('n', C.c_size_t)]


STEM_PARAM_FIELDS = ('conv0_w', 'conv0_b', 'b1_n1_w', 'b1_n1_b', 'b1_c1_w', 'b1_n2_w', 'b1_n2_b', 'b1_c2_w', 'b1_ds_w',
                     'b2_n1_w', 'b2_n1_b', 'b2_c1_w', 'b2_n2_w', 'b2_n2_b', 'b2_c2_w', 'b2_ds_w')


class NodeStemShape(C.Structure):
    _fields_ = [('n', C.c_int32), ('in_ch', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('filters', C.c_int32),
                ('eps', C.c_float)]


class NodeStemParams(C.Structure):          # node_stem_params and node_stem_grads share this layout
    _fields_ = [(k, C.c_void_p) for k in STEM_PARAM_FIELDS]


class NodeConvGeom(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('n', 'cin', 'cout', 'x_h', 'x_w', 'k', 'stride', 'pad')]


REDUCE_MEAN, REDUCE_SUM = 0, 1


class NodeHeadLoss(C.Structure):
    _fields_ = [('n', C.c_int32), ('c', C.c_int32), ('classes', C.c_int32), ('reduction', C.c_int32),
                ('pooled', C.c_void_p), ('weight', C.c_void_p), ('bias', C.c_void_p), ('target', C.c_void_p),
                ('logits', C.c_void_p), ('loss', C.c_void_p), ('stat', C.c_void_p), ('scratch', C.c_void_p)]


class NodeHeadLossGrad(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('grad_loss', 'grad_logits', 'd_logits', 'd_pooled', 'd_weight', 'd_bias')]


class NodeFlatSeg(C.Structure):
    _fields_ = [('y', C.c_void_p), ('y1', C.c_void_p), ('k', C.c_void_p * 7), ('n', C.c_size_t)]


class NodeFlatSolve(C.Structure):
    _fields_ = [('nseg', C.c_int32), ('has_scalar', C.c_int32), ('seg', NodeFlatSeg * 3), ('rtol', C.c_float), ('atol', C.c_float),
                ('tsign', C.c_float), ('n_targets', C.c_int32), ('ws', C.c_void_p), ('ws_bytes', C.c_size_t)]


class NodeFlatStatus(C.Structure):
    _fields_ = [('done', C.c_int32), ('status', C.c_int32), ('steps', C.c_int32), ('accepted', C.c_int32), ('rejected', C.c_int32),
                ('t', C.c_double), ('dt', C.c_double), ('first_dt', C.c_double), ('scalar', C.c_float)]


FLAT_F0, FLAT_PROBE = -1, -2


class NodeHipError(RuntimeError):
    def __init__(self, code, message):
        self.code = code
        super().__init__('libnode_hip: %s (%d): %s' % (ERRORS.get(code, '?'), code, message))


_lib = None


def load():
    """Load libnode_hip.so and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_DIAG_PATH if os.environ.get('NODE_HIP_DIAG', '0') not in ('', '0') else LIB_PATH
    if os.environ.get('NODE_HIP_LIB_AB'):          # tools: A/B of two builds of the library on one box (tools/ab_lib.sh)
        path = os.environ['NODE_HIP_LIB_AB']
    if not os.path.exists(path):
        raise RuntimeError(
            '%s is not built (%s). Run `python neural-ode-features_amd/build.py%s` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.'
            % (os.path.basename(path), path, ' --diag' if path == LIB_DIAG_PATH else ''))
    lib = C.CDLL(path)
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    P = C.POINTER
    lib.node_abi_version.restype = i32
    lib.node_abi_version.argtypes = []
    lib.node_last_error.restype = C.c_char_p
    lib.node_last_error.argtypes = []
    lib.node_param_count.restype = sz
    lib.node_param_count.argtypes = [P(NodeShape)]
    lib.node_workspace_bytes.restype = sz
    lib.node_workspace_bytes.argtypes = [P(NodeShape), i32, i32, i32]
    lib.node_solve_is_resident.restype = i32
    lib.node_solve_is_resident.argtypes = [P(NodeShape)]
    lib.node_odefunc_fwd.restype = i32
    lib.node_odefunc_fwd.argtypes = [P(NodeShape), P(NodeParams), f32, vp, vp, vp, sz, vp]
    lib.node_odefunc_vjp.restype = i32
    lib.node_odefunc_vjp.argtypes = [P(NodeShape), P(NodeParams), f32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.node_solve_fwd.restype = i32
    lib.node_solve_fwd.argtypes = [P(NodeShape), P(NodeParams), vp, P(C.c_float), i32, f32, f32, i32,
                                   P(NodeSolveOpts), vp, P(NodeStats), vp, sz, vp]
    lib.node_solve_adjoint.restype = i32
    lib.node_solve_adjoint.argtypes = [P(NodeShape), P(NodeParams), vp, vp, P(C.c_float), i32, f32, f32, i32,
                                       P(NodeSolveOpts), vp, vp, vp, P(NodeStats), vp, sz, vp]
    lib.node_backprop_workspace_bytes.restype = sz
    lib.node_backprop_workspace_bytes.argtypes = [P(NodeShape), i32, i32, i32]
    lib.node_solve_backprop.restype = i32
    lib.node_solve_backprop.argtypes = [P(NodeShape), P(NodeParams), vp, P(C.c_float), i32, P(C.c_double), i32, f32, f32, i32,
                                        vp, vp, vp, vp, sz, vp]
    lib.node_head_fwd.restype = i32
    lib.node_head_fwd.argtypes = [P(NodeShape), vp, vp, vp, vp, vp, vp, vp]
    lib.node_head_bwd.restype = i32
    lib.node_head_bwd.argtypes = [P(NodeShape), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.node_gn_relu_fwd.restype = i32
    lib.node_gn_relu_fwd.argtypes = [P(NodeShape), vp, vp, vp, i32, vp, vp, vp]
    lib.node_gn_relu_bwd.restype = i32
    lib.node_gn_relu_bwd.argtypes = [P(NodeShape), vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.node_sgd_step.restype = i32
    lib.node_sgd_step.argtypes = [P(NodeSgdTensor), i32, f32, f32, f32, f32, vp, vp]
    lib.node_profile_begin.restype = i32
    lib.node_profile_begin.argtypes = []
    lib.node_profile_end.restype = i32
    lib.node_profile_end.argtypes = [P(NodeProfile)]
    lib.node_conv3x3_w4_workspace_bytes.restype = sz
    lib.node_conv3x3_w4_workspace_bytes.argtypes = [P(NodeShape)]
    lib.node_w4_pair_stats.restype = i32
    lib.node_w4_pair_stats.argtypes = [P(C.c_int32)]
    lib.node_conv3x3_w4.restype = i32
    lib.node_conv3x3_w4.argtypes = [P(NodeShape), vp, i32, vp, vp, vp, sz, vp]
    lib.node_w4_split3.restype = i32
    lib.node_w4_split3.argtypes = [vp, vp, sz, vp]
    lib.node_stem_workspace_bytes.restype = sz
    lib.node_stem_workspace_bytes.argtypes = [P(NodeStemShape)]
    lib.node_stem_fwd.restype = i32
    lib.node_stem_fwd.argtypes = [P(NodeStemShape), P(NodeStemParams), vp, vp, vp, sz, vp]
    lib.node_stem_bwd.restype = i32
    lib.node_stem_bwd.argtypes = [P(NodeStemShape), P(NodeStemParams), vp, vp, P(NodeStemParams), vp, sz, vp]
    lib.node_stem_conv_workspace_bytes.restype = sz
    lib.node_stem_conv_workspace_bytes.argtypes = [P(NodeConvGeom)]
    lib.node_stem_conv.restype = i32
    lib.node_stem_conv.argtypes = [P(NodeConvGeom), i32, vp, vp, vp, vp, vp, sz, vp]
    lib.node_head_loss_scratch_bytes.restype = sz
    lib.node_head_loss_scratch_bytes.argtypes = [i32]
    lib.node_head_loss_fwd.restype = i32
    lib.node_head_loss_fwd.argtypes = [P(NodeHeadLoss), vp]
    lib.node_head_loss_bwd.restype = i32
    lib.node_head_loss_bwd.argtypes = [P(NodeHeadLoss), P(NodeHeadLossGrad), vp]
    lib.node_flat_workspace_bytes.restype = sz
    lib.node_flat_workspace_bytes.argtypes = [i32]
    lib.node_flat_begin.restype = i32
    lib.node_flat_begin.argtypes = [P(NodeFlatSolve), C.c_double, P(C.c_double), C.c_double, i32, vp]
    lib.node_flat_stage.restype = i32
    lib.node_flat_stage.argtypes = [P(NodeFlatSolve), i32, i32, P(C.c_void_p), vp, vp]
    lib.node_flat_scalar.restype = i32
    lib.node_flat_scalar.argtypes = [P(NodeFlatSolve), i32, vp, f32, i32, vp]
    lib.node_flat_initial_step.restype = i32
    lib.node_flat_initial_step.argtypes = [P(NodeFlatSolve), i32, vp]
    lib.node_flat_finish_step.restype = i32
    lib.node_flat_finish_step.argtypes = [P(NodeFlatSolve), i32, vp, vp]
    lib.node_flat_status_read.restype = i32
    lib.node_flat_status_read.argtypes = [P(NodeFlatSolve), P(NodeFlatStatus), vp]
    ver = lib.node_abi_version()
    if ver != NODE_ABI_VERSION:
        raise RuntimeError('libnode_hip ABI %d != binding ABI %d' % (ver, NODE_ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NodeHipError(rc, load().node_last_error().decode('utf-8', 'replace'))
