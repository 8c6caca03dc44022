from .clip_tree import tree_model  # noqa: F401
