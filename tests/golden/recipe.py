"""Import-path shim: the weight recipe lives in benchkit/recipe.py (shared with bench.py); the golden generators and the
tests keep importing `recipe`."""
from benchkit.recipe import key_seed, recipe_state_dict  # noqa: F401
