from .dataset import Dataset, PlainExample, to_example
from .vocab import Vocab, VocabEntry
