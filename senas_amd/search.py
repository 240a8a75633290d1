"""``python -m senas_amd.search --config <yml>``: the search phase (experiments/search_arc.py) on the HIP path."""
from .run import search

if __name__ == "__main__":
    search()
