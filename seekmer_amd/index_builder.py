"""Index construction -- the `seekmer.index_builder` surface (reference:
seekmer/index_builder.py, seekmer/_index_builder.pyx) over the native builder
in libseekmer_host.so.  The builder reproduces ContigAssembler's tables bit
for bit because contig orientation fixes the id order inside every
equivalence-class tuple.
"""
import ctypes
import logging
import pathlib
import re

import numpy

from . import _native
from . import common

__all__ = ('add_subcommand_parser', 'run', 'build', 'load_exome', 'read_transcripts',
           'ContigAssembler')

_LOG = logging.getLogger(__name__)

EXON_DTYPE_FIELDS = ('transcript_id', 'gene_id', 'exon_number', 'chromosome', 'start', 'end',
                     'strand')


class ContigAssembler:
    """A K-mer index builder using a de Bruijn graph
    (seekmer/_index_builder.pyx:85-150)."""

    def __init__(self, n_threads=1):
        self.n_threads = n_threads

    def assemble(self, sequences):
        """list[bytes] -> (kmer_table, contigs, sequences, targets)"""
        offsets = numpy.zeros(len(sequences) + 1, dtype=numpy.int64)
        numpy.cumsum([len(s) for s in sequences], out=offsets[1:])
        pool = numpy.frombuffer(b''.join(sequences) + b'\0', dtype=numpy.uint8)
        return self.assemble_pooled(pool, offsets)

    def assemble_pooled(self, pool, offsets):
        host = _native.host()
        built = ctypes.c_void_p()
        offsets = numpy.ascontiguousarray(offsets, dtype=numpy.int64)
        code = host.skm_build_index(pool.ctypes.data, _native.ptr(offsets, _native.c_i64p),
                                    offsets.size - 1, self.n_threads, ctypes.byref(built))
        if code == _native.SKM_ERR_UNDEFINED:
            raise RuntimeError('the reference assembler has undefined behaviour on this input')
        _native.check_host(code, 'skm_build_index')
        try:
            sizes = (ctypes.c_int64 * 4)()
            host.skm_built_sizes(built, sizes)
            kmers = numpy.zeros(sizes[0], dtype=common.KMER_DTYPE)
            contigs = numpy.zeros(sizes[1], dtype=common.CONTIG_DTYPE)
            sequences = numpy.zeros(sizes[2], dtype='S1')
            targets = numpy.zeros(max(sizes[3], 0), dtype=common.TARGET_DTYPE)
            host.skm_built_copy(built, kmers.ctypes.data, contigs.ctypes.data,
                                sequences.ctypes.data, targets.ctypes.data)
        finally:
            host.skm_built_free(built)
        return kmers, contigs, sequences, targets


def build(transcript_ids, sequences, exome=None):
    """Build an index for the given transcriptome (seekmer/index_builder.py:94-114)."""
    if len(transcript_ids) == 0:
        raise ValueError('no transcripts found')
    transcriptome, exome = _compile_omics(transcript_ids, sequences, exome)
    kmer_table, contigs, pooled, targets = ContigAssembler().assemble(sequences)
    return common.KMerIndex(kmer_table, contigs, pooled, targets, transcriptome, exome)


def build_pooled(transcript_ids, pool, offsets):
    """`build` for a transcriptome that is already one pooled byte array."""
    lengths = numpy.diff(offsets).astype('f8')
    id_length = max(len(i) for i in transcript_ids)
    transcriptome = numpy.zeros(len(transcript_ids),
                                dtype=[('transcript_id', 'S%d' % id_length),
                                       ('gene_id', 'S1'), ('length', 'f8')])
    transcriptome['transcript_id'] = transcript_ids
    transcriptome['length'] = lengths
    kmer_table, contigs, pooled, targets = ContigAssembler().assemble_pooled(pool, offsets)
    return common.KMerIndex(kmer_table, contigs, pooled, targets, transcriptome,
                            _empty_exome())


def _empty_exome():
    return numpy.zeros(0, dtype=[('transcript_id', 'S1'), ('gene_id', 'S1'), ('exon_number', 'i4'),
                                 ('chromosome', 'S1'), ('start', 'i4'), ('end', 'i4'),
                                 ('strand', '?')])


_ATTRIBUTE = {name: re.compile(name.encode() + rb' "([^"]+)"')
              for name in ('transcript_id', 'gene_id', 'exon_number')}


def load_exome(path):
    """Exon records of a GTF file as a record array with the reference's fields
    (seekmer/index_builder.py:117-170).  Parsed directly: the reference's pandas
    expression mis-parses under pandas >= 2 (SURVEY.md finding 7)."""
    rows = []
    with common.decompress_and_open(pathlib.Path(path)) as file:
        for line in file:
            if line[:1] == b'#':
                continue
            fields = line.rstrip(b'\n').split(b'\t')
            if len(fields) < 9 or fields[2] != b'exon':
                continue
            attributes = fields[8]
            values = {}
            for name, pattern in _ATTRIBUTE.items():
                match = pattern.search(attributes)
                values[name] = match.group(1) if match else b''
            try:
                exon_number = int(values['exon_number'])
            except ValueError:
                exon_number = 1                                   # ERCC GTFs, :142-144
            rows.append((values['transcript_id'], values['gene_id'], exon_number, fields[0],
                         int(fields[3]) - 1, int(fields[4]), fields[6] == b'+'))
    rows.sort(key=lambda r: (r[3], r[1], r[0], r[2]))
    widths = [max([len(r[i]) for r in rows] or [1]) for i in (0, 1, 3)]
    dtype = [('transcript_id', 'S%d' % widths[0]), ('gene_id', 'S%d' % widths[1]),
             ('exon_number', 'i4'), ('chromosome', 'S%d' % widths[2]), ('start', 'i4'),
             ('end', 'i4'), ('strand', '?')]
    return numpy.rec.fromrecords(rows, dtype=dtype) if rows else numpy.recarray(0, dtype=dtype)


def read_transcripts(fasta):
    """(ids, sequences) of a cDNA FASTA; id = first token up to the first '.'
    (seekmer/index_builder.py:173-184)."""
    transcript_ids, sequences = [], []
    for id_, seq in common.read_fasta(fasta):
        transcript_ids.append(id_.split()[0].split(b'.')[0])
        sequences.append(seq)
    return transcript_ids, sequences


def _compile_omics(transcript_ids, sequences, exome):
    """Transcript table {transcript_id, gene_id, length f8}
    (seekmer/index_builder.py:213-231)."""
    id_length = max(len(id_) for id_ in transcript_ids)
    ids = numpy.asarray(transcript_ids, dtype='S{}'.format(id_length))
    gene_dtype = 'S1'
    if exome is not None and len(exome):
        exome = exome[numpy.isin(exome['transcript_id'], ids)]
        exome = numpy.sort(exome)
        gene_dtype = exome.dtype.fields['gene_id'][0]
    else:
        exome = _empty_exome()
    transcriptome = numpy.zeros(len(ids), dtype=[('transcript_id', ids.dtype),
                                                 ('gene_id', gene_dtype), ('length', 'f8')])
    transcriptome['transcript_id'] = ids
    if len(exome):
        where = numpy.searchsorted(exome['transcript_id'], ids)
        picked = numpy.take(exome, where, mode='clip')
        transcriptome['gene_id'] = numpy.where(picked['transcript_id'] != ids, b'', picked['gene_id'])
    transcriptome['length'] = [len(seq) for seq in sequences]
    return transcriptome, exome


def run(fasta_path, gtf_path, index_path, use_transcriptome=False, **__):
    """Generate an index file (seekmer/index_builder.py:65-91; transcriptome
    FASTA input only -- genome extraction is outside the infer hot path)."""
    _LOG.info('Building index')
    exome = load_exome(gtf_path) if gtf_path is not None else None
    if not use_transcriptome:
        raise NotImplementedError('genomic FASTA input is out of scope: pass -t with a cDNA FASTA')
    transcript_ids, sequences = read_transcripts(fasta_path)
    index = build(transcript_ids, sequences, exome)
    index.save(index_path)


def add_subcommand_parser(subparsers):
    """seekmer/index_builder.py:44-62"""
    parser = subparsers.add_parser('index', help='build a Seekmer index')
    parser.add_argument('-t', '--transcriptome', action='store_true', dest='use_transcriptome',
                        help='use transcriptomic sequences instead of genomic sequences')
    parser.add_argument('fasta_path', type=pathlib.Path, metavar='fasta',
                        help='specify a transcriptome sequence file')
    parser.add_argument('gtf_path', type=pathlib.Path, metavar='gtf',
                        help='specify a transcript annotation file')
    parser.add_argument('index_path', type=pathlib.Path, metavar='index',
                        help='specify an output index file')
