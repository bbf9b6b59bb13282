#!/usr/bin/env python3
"""CLI (reference: seekmer/__main__.py:13-71): `seekmer_amd [--debug] {index,infer,impute} ...`"""
import argparse
import logging
import sys

from . import impute
from . import index_builder
from . import infer


def main(argv=None):
    parser = argparse.ArgumentParser(prog='seekmer_amd', description='A fast RNA-seq tool',
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('-v', '--version', action='version', version='Seekmer 2019.0.0 (MI355X)')
    parser.add_argument('--debug', action='store_true', help='enable debugging messages')
    subparsers = parser.add_subparsers(title='subcommand', dest='subcommand')
    index_builder.add_subcommand_parser(subparsers)
    infer.add_subcommand_parser(subparsers)
    impute.add_subcommand_parser(subparsers)
    opts = vars(parser.parse_args(argv))
    logging.basicConfig(level=logging.DEBUG if opts['debug'] else logging.INFO,
                        format='%(levelname)-5s %(asctime)s %(name)s: %(message)s',
                        datefmt='%Y-%m-%d %H:%M:%S', stream=sys.stderr)
    if opts['subcommand'] == 'index':
        index_builder.run(**opts)
    elif opts['subcommand'] == 'infer':
        infer.run(**opts)
    elif opts['subcommand'] == 'impute':
        impute.run(**opts)
    else:
        parser.print_help()
    return 0


if __name__ == '__main__':
    sys.exit(main())
