"""The bench line the driver parses: the committed lines (profiles/r03_bench.json, written by the default
`python bench.py` on the MI355X box; profiles/r02_bench*.json from the round before) carry every field
of the contract, with the metric and the workloads BASELINE.json names, and their numbers hang together
(value = units / time, roofline.frac = achieved / peak, achieved = algorithmic bytes / launch time).
The default line also carries the legs from outside the GPU (`e2e`) and a few timed steps of the other
single-GPU configurations of BASELINE.json (`other_configs`)."""
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


@pytest.mark.parametrize('name, config', [('r04_bench.json', 1), ('r03_bench.json', 1), ('r02_bench.json', 1),
                                          ('r02_bench_config3.json', 3), ('r02_bench_config4.json', 4)])
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
        # (up to the middle of round 4 the counters showed MORE than the algorithmic bytes -- wasted
        # re-reads; since the signatures of the first-hit roll they show less: the product proves most
        # of the roll's k-mers absent without touching the buckets the reference's probe reads)
        assert 0.5 * roofline['algorithmic_bytes_per_launch'] < roofline['traffic'] < 3 * roofline['algorithmic_bytes_per_launch']
        assert 'profiles/' in roofline['traffic_source']
    cpu = line['cpu_baseline']
    for field in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert field in cpu, field
    assert cpu['kind'] in ('port', 'reference') and cpu['unit'] == line['unit'] and cpu['value'] < line['value']
    phases = line['config']['phase_ms']
    assert sum(phases.values()) <= line['ms_per_step'] * 1.001


def test_default_line_reports_the_legs_from_outside_the_gpu():
    line = _line('r03_bench.json')
    e2e = line['e2e']
    pcie, fastq = e2e['pcie_inclusive'], e2e['fastq_inclusive']
    assert pcie['value'] < line['value'] and pcie['unit'] == 'pairs/s'
    assert fastq['value'] < pcie['value']
    # the host-array leg cannot beat the link: its bytes over its time stay under PCIe 5 x16
    assert pcie['GBps_over_pcie'] < 64.0 and pcie['ascii']['GBps_over_pcie'] < 64.0
    assert pcie['bytes_per_pair'] == 64.0                          # 2 x 100 bases as 2-bit code words
    assert pcie['first_pass'] <= pcie['value'] and fastq['first_pass'] <= fastq['value']
    # one pass over the text: every piece accepted as guessed, every read seen once
    assert fastq['reader']['reparsed'] == 0 and fastq['reader']['units'] == line['config']['units_per_gpu']
    assert fastq['reader']['reads'] == 2 * line['config']['units_per_gpu']
    assert fastq['value'] > fastq['two_pass_ascii']['value'] and pcie['value'] > pcie['ascii']['value']


def test_default_line_times_the_other_single_gpu_configs():
    line = _line('r03_bench.json')
    baseline = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    others = line['other_configs']
    assert sorted(others) == ['configs[3]', 'configs[4]'] and len(baseline['configs']) > 4
    for name, sub in others.items():
        for field in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'workload', 'phase_ms', 'roofline'):
            assert field in sub, (name, field)
        assert sub['workload'].startswith(name) and sub['steps'] >= 3 and sub['warmup'] >= 1
        assert sum(sub['phase_ms'].values()) <= sub['ms_per_step'] * 1.001
        roofline = sub['roofline']
        assert roofline['frac'] == pytest.approx(roofline['achieved'] / roofline['peak'], rel=1e-9)
        assert roofline['achieved'] == pytest.approx(
            roofline['algorithmic_bytes_per_launch'] / (roofline['launch_ms'] * 1e-3) / 1e9, rel=1e-6)
    third, fourth = others['configs[3]'], others['configs[4]']
    assert third['unit'] == 'reads/s' and '50000000 150bp single-end' in third['workload']
    assert third['value'] == pytest.approx(50_000_000 / (third['ms_per_step'] * 1e-3), rel=1e-6)
    assert fourth['unit'] == 'pairs/s' and fourth['bootstraps'] == 100 and fourth['bootstraps_per_s'] > 0
    assert fourth['value'] == pytest.approx(20_000_000 / (fourth['ms_per_step'] * 1e-3), rel=1e-6)


def test_round4_line_checks_its_own_parity_and_times_the_north_star_size():
    """profiles/r04_bench.json: the default run compares the counting build's table with the production launch's at every
    benched size and the oracle's table with the HIP path's on the cpu_baseline sample (a mismatch fails the run), times
    BASELINE.json north_star's 50 M 2x100 pairs on one GPU, and reports what one cold process sees from FASTQ text."""
    line = _line('r04_bench.json')
    parity = line['parity_checked']
    assert parity['counting_vs_production']['identical'] is True and parity['counting_vs_production']['units'] == 10_000_000
    sample = parity['cpu_sample_vs_hip']
    assert sample['tables_identical'] is True and sample['units'] >= 1_000_000 and sample['tpm_max_rel_diff'] < 1e-4
    others = line['other_configs']
    assert sorted(others) == ['configs[3]', 'configs[4]', 'north_star: 50 M 2x100 pairs']
    for sub in others.values():
        assert sub['parity_checked']['counting_vs_production']['identical'] is True
    star = others['north_star: 50 M 2x100 pairs']
    assert star['unit'] == 'pairs/s' and '50000000 2x100bp' in star['workload'] and star['steps'] >= 2
    assert star['value'] == pytest.approx(50_000_000 / (star['ms_per_step'] * 1e-3), rel=1e-6)
    assert star['parity_checked']['counting_vs_production']['units'] == 50_000_000
    roofline = line['roofline']
    assert roofline['traffic'] is not None and 'r04_pmc_map.json' in roofline['traffic_source']
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r04_pmc_map.json')))
    assert roofline['traffic'] == pmc['derived']['hbm_traffic_bytes'] and len(pmc['source_hash']) == 16
    cold = line['e2e']['fastq_inclusive']['cold_process']
    assert cold['unit'] == 'pairs/s' and cold['without_prefault']['value'] <= cold['value'] < line['e2e']['fastq_inclusive']['parse_only']
    assert line['e2e']['fastq_inclusive']['first_pass'] <= line['e2e']['fastq_inclusive']['value']
