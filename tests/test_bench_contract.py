"""The bench line the driver parses: the committed lines of this round (profiles/r02_bench*.json, written
by `python bench.py [--config N]` on the MI355X box) carry every field of the contract, with the
metric and the workloads BASELINE.json names, and their numbers hang together (value = units / time,
roofline.frac = achieved / peak, achieved = algorithmic bytes / launch time)."""
import json
import os

import pytest

from conftest import ROOT

FIELDS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
          'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _line(name):
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        pytest.skip('%s not committed' % name)
    return json.load(open(path))


@pytest.mark.parametrize('name, config', [('r02_bench.json', 1), ('r02_bench_config3.json', 3),
                                          ('r02_bench_config4.json', 4)])
def test_committed_bench_lines_follow_the_contract(name, config):
    line = _line(name)
    baseline = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    for field in FIELDS:
        assert field in line, field
    assert line['higher_is_better'] is True and line['scaling'] == 'weak' and line['data'] == 'synthetic'
    assert line['n_gpus'] == 1 and line['vs_baseline'] is None
    assert line['unit'] in ('pairs/s', 'reads/s')
    assert line['metric'].split(' (')[0] in baseline['metric'] or config == 3      # (-s: single reads per second)
    assert 'model' not in line['config'] and line['config']['workload'].startswith('configs[%d]' % config)
    assert len(baseline['configs']) > config
    units = line['config']['units_per_gpu'] * line['n_gpus']
    assert line['value'] == pytest.approx(units / (line['ms_per_step'] * 1e-3), rel=1e-6)
    roofline = line['roofline']
    for field in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert field in roofline, field
    assert roofline['bound'] == 'hbm' and roofline['unit'] == 'GB/s' and roofline['peak'] == 8000.0
    assert roofline['frac'] == pytest.approx(roofline['achieved'] / roofline['peak'], rel=1e-9)
    assert roofline['achieved'] == pytest.approx(
        roofline['algorithmic_bytes_per_launch'] / (roofline['launch_ms'] * 1e-3) / 1e9, rel=1e-6)
    assert 0 < roofline['frac'] < 1
    if roofline['traffic'] is not None:          # (quoted from the PMC passes of the same build)
        assert roofline['traffic'] >= roofline['algorithmic_bytes_per_launch']
        assert 'profiles/' in roofline['traffic_source']
    cpu = line['cpu_baseline']
    for field in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert field in cpu, field
    assert cpu['kind'] in ('port', 'reference') and cpu['unit'] == line['unit'] and cpu['value'] < line['value']
    phases = line['config']['phase_ms']
    assert sum(phases.values()) <= line['ms_per_step'] * 1.001


def test_default_line_reports_the_legs_from_outside_the_gpu():
    line = _line('r02_bench.json')
    e2e = line['e2e']
    assert e2e['pcie_inclusive']['value'] < line['value'] and e2e['pcie_inclusive']['unit'] == 'pairs/s'
    assert e2e['fastq_inclusive']['value'] < e2e['pcie_inclusive']['value']
    # the host-array leg cannot beat the link: its bytes over its time stay under PCIe 5 x16
    assert e2e['pcie_inclusive']['GBps_over_pcie'] < 64.0
