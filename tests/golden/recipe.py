"""Import-path shim: the weight recipe lives in benchkit/recipe.py (shared with bench.py); the golden generators and the
tests keep importing `recipe`.  Loaded BY FILE PATH, not as `benchkit.recipe`: the generators run with the reference checkout
first on sys.path and must not put this repository's root there (its `lib/` is a regular package and would shadow the
reference's namespace package `lib/` whatever the order)."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "benchkit", "recipe.py")
_spec = importlib.util.spec_from_file_location("_smap_benchkit_recipe", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
key_seed, recipe_state_dict = _mod.key_seed, _mod.recipe_state_dict
