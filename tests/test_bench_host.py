"""Host-side pieces of bench.py and of the Predictor's pipeline that need no GPU: the sysfs sensor sampler (fake amdgpu hwmon tree, and a machine without one)
and the huge-page advice on the result volume."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_card(root, name, pci, sclk_hz, power_uw, temp_mc):
    dev = root / 'pci' / pci
    hw = dev / 'hwmon' / 'hwmon3'
    hw.mkdir(parents=True)
    (hw / 'freq1_input').write_text(f'{sclk_hz}\n')
    (hw / 'power1_input').write_text(f'{power_uw}\n')
    (hw / 'power1_cap').write_text('1400000000\n')
    (hw / 'temp2_input').write_text(f'{temp_mc}\n')
    card = root / 'drm' / name
    card.mkdir(parents=True)
    os.symlink(dev, card / 'device')


def test_gpu_sensors_read_this_gpu_and_count_busy_neighbours(tmp_path, monkeypatch):
    import glob as _glob
    import bench
    _fake_card(tmp_path, 'card0', '0000:05:00.0', 2350000000, 1200000000, 51000)       # this process' GPU (bus 5)
    _fake_card(tmp_path, 'card8', '0000:15:00.0', 2400000000, 900000000, 60000)        # a busy neighbour
    _fake_card(tmp_path, 'card16', '0000:25:00.0', 96000000, 200000000, 40000)         # an idle one
    real_glob = _glob.glob
    monkeypatch.setattr(_glob, 'glob', lambda p, **kw: real_glob(p.replace('/sys/class/drm', str(tmp_path / 'drm')), **kw))
    monkeypatch.setattr(torch.cuda, 'get_device_properties', lambda dev: types.SimpleNamespace(pci_bus_id=5))
    with bench.GpuSensors(torch.device('cpu')) as s:
        time.sleep(0.2)
    out = s.summary()
    assert out['samples'] >= 2 and out['other_gpus_seen'] == 2
    assert out['sclk_mhz_min_mean_max'] == [2350.0, 2350.0, 2350.0] and out['socket_power_w_min_mean_max'][1] == 1200.0
    assert out['temp_c_min_mean_max'][1] == 51.0 and out['power_cap_w'] == 1400.0 and out['other_gpus_of_the_node_busy_mean'] == 1.0


def test_gpu_sensors_without_sysfs_report_nothing(monkeypatch):
    import bench

    def boom(dev):
        raise RuntimeError('no device')
    monkeypatch.setattr(torch.cuda, 'get_device_properties', boom)
    with bench.GpuSensors(torch.device('cpu')) as s:
        pass
    assert s.summary() is None


def test_hugepage_advice_is_harmless_on_any_host_tensor(monkeypatch):
    from elektronn3_amd import inference
    small = torch.empty(1024)
    inference._advise_hugepages(small)                      # below 64 MB: nothing happens
    big = torch.empty(80 << 20, dtype=torch.uint8)
    inference._advise_hugepages(big)                        # advised (or refused by the kernel): the tensor is an ordinary tensor afterwards
    big[:4096] = 7
    big[-4096:] = 9
    assert int(big[0]) == 7 and int(big[-1]) == 9
    monkeypatch.setenv('E3_PREDICTOR_NO_HUGEPAGES', '1')
    inference._advise_hugepages(torch.empty(80 << 20, dtype=torch.uint8))


def test_every_switch_of_the_gpu_switch_groups_is_read_somewhere():
    """A switch group that sets an environment variable nobody reads would silently test the default path: every E3_* name in tests/test_switches_gpu.py's
    groups and in README.md's switch list must occur in the library's or the package's sources."""
    import glob
    import re
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_switches_gpu as sw
    src = ''
    for pat in ('elektronn3_amd/csrc/*', 'elektronn3_amd/*.py', 'bench.py'):
        for f in glob.glob(os.path.join(ROOT, pat)):
            if os.path.isfile(f) and not f.endswith('.so'):
                src += open(f, errors='ignore').read()
    names = {k for env, _ in sw.GROUPS.values() for k in env}
    missing = sorted(n for n in names if n not in src)
    assert not missing, f'switches set by the test groups but read nowhere: {missing}'
    readme = open(os.path.join(ROOT, 'README.md')).read()
    start = readme.index('Switches for A/B runs')
    listed = set(re.findall(r'`(E3_[A-Z0-9_]+)', readme[start:start + 12000]))
    removed = readme[readme.index('Removed (alternatives'):][:1500] if 'Removed (alternatives' in readme else ''
    stale = sorted(n for n in listed if n not in src and n not in removed)
    assert not stale, f'switches documented in README.md but read nowhere: {stale}'
