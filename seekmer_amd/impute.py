"""Single-cell imputation -- the `seekmer.impute` surface (reference:
seekmer/impute.py:1-252) over the MI355X engine.  Every cell is mapped into its
own device-resident class table (`mapper.map_multiple_samples`), the cells'
fragment-length histograms are pooled, every cell is quantified once, a
cell-by-cell weight matrix is derived from the gene-level abundances, and every
cell is quantified again on the weighted blend of all cells' class tables.

What the engine changes: the blended problem has ONE class structure (the
concatenation of all cells' classes, impute.py:243-247) and only the class
counts differ from cell to cell (:248-252), so its two CSR views are built on
the GPU once and each cell's second round is `set_counts` + the EM kernels.
The arithmetic that decides results is the reference's, cited per function.
"""
import logging
import pathlib

import numpy

from . import common
from . import infer
from . import mapper

__all__ = ('add_subcommand_parser', 'run')

_LOG = logging.getLogger(__name__)


def run(index_path, output_path, fastq_paths, job_count, single_ended, debug, power,
        device=0, seed=None, **__):
    """The entrypoint of the imputation module (seekmer/impute.py:54-125).
    `seed` fixes the 2-means split of the weights (the reference leaves it to
    numpy's global generator)."""
    import pandas
    for path in fastq_paths:
        if not pathlib.Path(path).exists():
            raise ValueError(f'invalid FastQ file: {path}')
    try:
        output_path.mkdir(parents=True)
    except FileExistsError:
        _LOG.warning('The output folder exists. Overriding...')
    _LOG.info('Inferring transcript abundance')
    index = common.KMerIndex.load(index_path)
    _LOG.info('Mapping all reads')
    width = 1 if single_ended else 2
    groups = [tuple(fastq_paths[i:i + width]) for i in range(0, len(fastq_paths) - width + 1, width)]
    # (a cell's reads: plain files are parsed straight to the mapper's 2-bit codes by the thread that
    # maps the cell -- one pass, a third of the bytes over PCIe; compressed ones through the ASCII reader)
    feeders = [common.PackedReadFeeder(list(group), paired=not single_ended)
               if common.PackedReadFeeder.eligible(group) else common.NativeReadFeeder(list(group), paired=not single_ended)
               for group in groups]
    map_results = mapper.map_multiple_samples(index, feeders, job_count=job_count, debug=debug,
                                              device=device)
    _LOG.info('Mapped all reads.')
    pool_fragment_lengths(map_results)
    summaries = [result.summarize() for result in map_results]
    _LOG.info('First round quantification...')
    base = numpy.asarray([infer.quantify(summary) for summary in summaries])
    if power is not None:
        _LOG.info('Weighting cells.')
        weight = cell_weights(index, base, output_path, seed=seed)
        weight **= power
        _LOG.info('Second round quantification...')
        columns = requantify_blend(summaries, weight, device=device)
    else:
        columns = list(base)
    names = [str(group[0]) for group in groups]          # one column per cell, named by its first file
    table = pandas.DataFrame(dict(zip(names, columns)),
                             index=numpy.char.decode(index.transcripts['transcript_id']))
    _LOG.info('Writing results to %s...', output_path)
    table.to_csv(output_path / 'tpm.csv')
    return table


def pool_fragment_lengths(map_results):
    """Single cells usually share one sequencing batch: every cell gets the sum
    of all cells' fragment-length counts (seekmer/impute.py:128-146).  The
    histograms live on the GPU; each one receives what the others counted."""
    counts = [result.fragment_length_counts for result in map_results]
    total = numpy.sum(counts, axis=0, dtype=numpy.int64) if counts else None
    for result, own in zip(map_results, counts):
        result.merge_fragment_lengths(total - own)


def gene_matrix(index, tpm_matrix):
    """Cell-by-gene abundance, truncated to integers as the reference's i8
    matrix does on assignment, genes named b'' dropped (impute.py:205-211)."""
    genes, gene_of_transcript = numpy.unique(index.transcripts['gene_id'], return_inverse=True)
    tpm_matrix = numpy.asarray(tpm_matrix, dtype='f8')
    # the transcripts of a gene side by side (stable: in transcript order), so that each gene is
    # one column slice and its row sums are numpy's own, as in the reference's masked .sum(axis=1)
    order = numpy.argsort(gene_of_transcript, kind='stable')
    by_gene = tpm_matrix[:, order]
    bounds = numpy.searchsorted(gene_of_transcript[order], numpy.arange(len(genes) + 1))
    matrix = numpy.zeros((len(tpm_matrix), len(genes)), dtype='i8')
    for gene in range(len(genes)):
        block = numpy.ascontiguousarray(by_gene[:, bounds[gene]:bounds[gene + 1]])
        matrix[:, gene] = block.sum(axis=1)          # (float -> i8 on assignment: truncation)
    named = genes != b''
    return matrix[:, named], genes[named]


def cell_weights(index, tpm_matrix, output_path=None, seed=None):
    """Cell-by-cell weights (seekmer/impute.py:187-226): Pearson correlation of
    the integer gene-level abundances; the off-diagonal, defined correlations
    are split in two by 1-D 2-means and only those of the upper cluster keep
    their value, everything else (and every NaN) becomes 0."""
    import pandas
    import sklearn.cluster
    matrix, genes = gene_matrix(index, tpm_matrix)
    if output_path is not None:
        pandas.DataFrame(matrix.T, index=numpy.char.decode(genes)).to_csv(
            output_path / 'initial_gene_table.csv')
    with numpy.errstate(invalid='ignore', divide='ignore'):
        weights = numpy.corrcoef(matrix)
    weights = numpy.atleast_2d(weights)
    informative = weights[(weights == weights) & (weights != 1.0)]
    split = sklearn.cluster.KMeans(2, random_state=seed)
    split.fit(informative[:, None])
    weights[weights != weights] = 0.0
    upper = split.predict(weights.reshape(-1, 1)) == split.cluster_centers_.argmax()
    weights = numpy.where(upper.reshape(weights.shape), weights, 0.0)
    if output_path is not None:
        pandas.DataFrame(weights).to_csv(output_path / 'weight.csv')
    return weights


def blend(summaries, weight):
    """The blended problem of seekmer/impute.py:229-252 in CSR form: (offsets,
    targets) = all cells' classes one after the other; counts[i] = for cell i
    the concatenation over cells j of count_j * weight[i, j] * total_i /
    total_j (totals = sums of the cells' own class counts)."""
    for summary in summaries:
        if summary.class_count.size == 0:
            raise ValueError('a cell without aligned reads cannot be blended')   # max() of nothing, :241
    sizes = [numpy.bincount(s.class_map[0].astype(numpy.int64), minlength=s.class_count.size)
             for s in summaries]
    offsets = numpy.zeros(sum(s.size for s in sizes) + 1, dtype=numpy.int64)
    numpy.cumsum(numpy.concatenate(sizes), out=offsets[1:])
    targets = numpy.concatenate([s.class_map[1] for s in summaries]).astype(numpy.int32)
    own = [s.class_count for s in summaries]
    counts = []
    for i, summary in enumerate(summaries):
        total = summary.class_count.sum()
        counts.append(numpy.concatenate([c * w * total / c.sum() for c, w in zip(own, weight[i, :])]))
    return offsets, targets, counts


def requantify_blend(summaries, weight, device=0):
    """quantify() of every cell's blended table (seekmer/impute.py:101-108):
    one quantification handle for the shared class structure, the cell's counts
    swapped in for each run."""
    offsets, targets, counts = blend(summaries, weight)
    n_tx = summaries[0].effective_lengths.size
    handle = infer._QuantHandle.from_csr(n_tx, offsets, targets, counts[0], device)
    columns = []
    try:
        for summary, cell_counts in zip(summaries, counts):
            lengths = summary.effective_lengths.astype('f8')
            x = numpy.ones(n_tx, dtype='f8') / lengths                  # infer.py:116-119
            x /= x.sum()
            handle.set_counts(cell_counts)
            x, _ = handle.em(x, lengths)
            columns.append(infer._tpm(x))
    finally:
        handle.close()
    return columns


def add_subcommand_parser(subparsers):
    """Add an impute command to the subparsers (seekmer/impute.py:18-51)."""
    parser = subparsers.add_parser(
        'impute', help='impute transcript abundance for single-cell data',
        epilog='Demultiplex the read files first and list them as arguments: every two files are one '
               'cell; with "-s" (single-ended reads) every file is one cell.')
    parser.add_argument('index_path', type=pathlib.Path, metavar='index',
                        help='specify a Seekmer index file')
    parser.add_argument('output_path', type=pathlib.Path, metavar='output',
                        help='specify a output folder')
    parser.add_argument('fastq_paths', type=pathlib.Path, metavar='fastq', nargs='+',
                        help='specify a FASTQ read file')
    parser.add_argument('-j', '--jobs', type=int, dest='job_count', metavar='N', default=1,
                        help='specify the maximum parallel job number')
    parser.add_argument('-p', '--power', type=int, dest='power', metavar='P', default=16,
                        help='specify the power of the weight matrix')
    parser.add_argument('-m', '--save-readmap', action='store_true', dest='save_readmap',
                        help='output an readmap file')
    parser.add_argument('-s', '--single-ended', action='store_true', dest='single_ended',
                        help='specify whether the reads are single-ended')
    parser.add_argument('--device', type=int, default=0, help='GPU ordinal (default 0)')
    parser.add_argument('--seed', type=int, default=None,
                        help='seed of the 2-means split of the cell weights (default: random)')
