"""Command line of train.py (reference: modules/config/Parser.py:3-12): positional dataroot,
-n/--numepochs, -r/--resume.  Unknown arguments are ignored instead of aborting the import."""
import argparse
import sys

_parser = argparse.ArgumentParser(add_help=False)
_parser.add_argument('-n', '--numepochs', type=int, dest='numepochs', default=10)
_parser.add_argument('-r', '--resume', type=int, dest='lastiter', default=0)
_parser.add_argument('rest', nargs='*')
try:
    options, _unknown = _parser.parse_known_args(sys.argv[1:])
    args = [a for a in options.rest if not a.startswith('-') and not a.endswith('.py')]
except SystemExit:                       # pragma: no cover - malformed argv of a host program
    options, args = argparse.Namespace(numepochs=10, lastiter=0), []
if 'pytest' in sys.modules or any('pytest' in a for a in sys.argv[:1]):
    args = []
