"""CPU-only: the C-ABI library builds, loads and exports every symbol include/e3unet.h declares, with the argument
counts the ctypes binding uses.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'e3unet.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'^\s*#.*$', '', src, flags=re.M)
    fns = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(e3_\w+)\s*\(([^;{}]*?)\)\s*;', src, flags=re.S):
        name, args = m.group(2), m.group(3).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        fns[name] = n
    return fns


def test_header_declares_something():
    fns = header_functions()
    assert 'e3_unet_forward' in fns and 'e3_conv3d_fwd' in fns and len(fns) >= 30


def test_library_exports_all_header_symbols():
    from elektronn3_amd import _lib
    lib = _lib.load()
    fns = header_functions()
    for name in fns:
        assert hasattr(lib, name), f'{name} declared in include/e3unet.h but not exported by libe3unet.so'
    assert lib.e3_version().decode().startswith('e3unet')


def test_binding_matches_header_arity():
    from elektronn3_amd import _lib
    fns = header_functions()
    assert set(_lib.EXPORTED_SYMBOLS) == set(fns), set(_lib.EXPORTED_SYMBOLS) ^ set(fns)
    for name, (_, args) in _lib._SIG.items():
        assert len(args) == fns[name], (name, len(args), fns[name])


def test_integration_doc_struct_matches_the_binding():
    """The ctypes `UNetCfg` a maintainer would copy from INTEGRATION.md has the fields, order and types of the real binding (a stale,
    shorter struct makes the library read `act_slope` past its end) and the header's e3_unet_cfg."""
    from elektronn3_amd import _lib
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = doc[doc.index('class UNetCfg(ctypes.Structure)'):doc.index('plan = ctypes.c_void_p()')]
    fields = re.findall(r"\('(\w+)',\s*ctypes\.(c_\w+)\)", block)
    want = [(n, t.__name__) for n, t in _lib.UNetCfg._fields_]
    norm = lambda t: {'c_int': 'c_int32', 'c_uint': 'c_uint32'}.get(t, t)      # (ctypes aliases c_int32 to c_int on this platform)
    assert [(n, norm(t)) for n, t in fields] == [(n, norm(t)) for n, t in want], (fields, want)
    hdr = open(os.path.join(ROOT, 'include', 'e3unet.h')).read()
    struct = hdr[hdr.index('typedef struct e3_unet_cfg {'):hdr.index('} e3_unet_cfg;')]
    assert re.findall(r'^\s*(?:u?int32_t|float)\s+(\w+);', struct, flags=re.M) == [n for n, _ in want]
    call = re.search(r'cfg = UNetCfg\(([^)]*)\)', doc).group(1)
    assert len([a for a in call.split(',') if a.strip()]) == len(want)


@pytest.mark.parametrize('fixture,cfg_args', [('unet_nb4_sf8_planar01.npz', (1, 2, 4, 8, 0b0011, 1, 1e-5, 1)),
                                              ('unet_nb3_sf8_planar0_sparsenorm.npz', (1, 2, 3, 8, 0b001, 1, 1e-5, 0)),   # full_norm=False
                                              ('unet_nb2_sf8_nonorm.npz', (1, 2, 2, 8, 0, 0, 1e-5, 1))])                  # normalization='none'
def test_plan_param_table_matches_reference_state_dict_names(fixture, cfg_args):
    """Host logic only (no GPU): the plan's parameter table uses the reference's state_dict keys (golden fixtures sd0 of an
    n_blocks=4, planar_blocks=(0,1) model, a full_norm=False model and a normalization='none' model)."""
    import numpy as np
    from elektronn3_amd import _lib
    from helpers import load_npz, sub
    g = load_npz(fixture)
    sd = sub(g, 'sd0')
    lib = _lib.load()
    cfg = _lib.UNetCfg(*cfg_args)
    plan = ctypes.c_void_p()
    _lib.check(lib.e3_unet_plan_create(ctypes.byref(cfg), ctypes.byref(plan)))
    try:
        names = []
        for i in range(lib.e3_unet_param_count(plan)):
            buf = ctypes.create_string_buffer(128)
            numel, kind = ctypes.c_int64(), ctypes.c_int()
            _lib.check(lib.e3_unet_param_info(plan, i, buf, 128, ctypes.byref(numel), ctypes.byref(kind)))
            name = buf.value.decode()
            names.append(name)
            assert name in sd, name
            assert int(np.prod(sd[name].shape)) == numel.value, name
        expected = {k for k in sd if not k.endswith('num_batches_tracked')}
        assert set(names) == expected
        assert lib.e3_unet_bn_count(plan) == sum(1 for k in sd if k.endswith('running_mean'))
        saved, scratch = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(lib.e3_unet_sizes(plan, 2, 8, 32, 32, 1, ctypes.byref(saved), ctypes.byref(scratch)))
        assert saved.value > 0 and scratch.value > 0
    finally:
        lib.e3_unet_plan_destroy(plan)


def test_unsupported_config_is_reported():
    from elektronn3_amd import _lib
    lib = _lib.load()
    cfg = _lib.UNetCfg(1, 2, 3, 12, 0, 1, 1e-5)   # start_filts not a multiple of 8
    plan = ctypes.c_void_p()
    with pytest.raises(NotImplementedError):
        _lib.check(lib.e3_unet_plan_create(ctypes.byref(cfg), ctypes.byref(plan)))


def test_optimizer_and_criterion_host_logic_fails_loudly_without_a_gpu():
    """Host side of elektronn3_amd.optim.SWA / AdamW and the criterion: the reference wrapper's argument validation (training/swa.py:88-112)
    and NO CPU path -- CPU tensors raise instead of silently computing somewhere else."""
    import pytest
    import torch
    from elektronn3_amd.loss import CombinedCEDiceLoss
    from elektronn3_amd.optim import SWA, AdamW
    p = torch.nn.Parameter(torch.randn(5))
    with pytest.raises(ValueError):
        SWA(torch.optim.SGD([p], lr=0.1), swa_start=-1, swa_freq=2)
    with pytest.raises(ValueError):
        SWA(torch.optim.SGD([p], lr=0.1), swa_start=1, swa_freq=0)
    with pytest.raises(ValueError):
        SWA(torch.optim.SGD([p], lr=0.1), swa_start=1, swa_freq=2, swa_lr=-0.1)
    with pytest.warns(UserWarning):
        manual = SWA(torch.optim.SGD([p], lr=0.1), swa_start=3)          # only one of the two given: manual mode
    assert not manual._auto_mode and manual.swa_start is None and manual.param_groups[0]['n_avg'] == 0
    auto = SWA(torch.optim.SGD([p], lr=0.1), swa_start=1, swa_freq=1)
    p.grad = torch.ones(5)
    auto.step()                                      # step 1: not yet averaging (steps > swa_start)
    assert auto.param_groups[0]['step_counter'] == 1 and 'swa_buffer' not in auto.state[p]
    with pytest.raises(RuntimeError):
        auto.step()                                  # step 2 averages: CPU parameters have no path
    with pytest.raises(RuntimeError):
        opt = AdamW([p]); opt.step()
    with pytest.raises(ValueError):
        CombinedCEDiceLoss(weight=[0.3, 0.7])(torch.randn(1, 2, 4, 4, 4), torch.zeros(1, 4, 4, 4, dtype=torch.int64))
