"""Docstring inheritance for penalty classes - the names user code imports from the reference's `matcouply._doc_utils`
(`copy_ancestor_docstring`, `InheritableDocstrings`; used e.g. by the reference's examples/plot_custom_penalty.py:213-231).

Own design: the decorator only MARKS a method; the metaclass of the penalty tree (a subclass of ABCMeta, so abstract
methods keep working) fills the docstring of every marked method from the first ancestor that documents it, once the
class exists.  Outside such a class the decorator is a harmless marker - the method simply keeps no docstring.
"""
from abc import ABCMeta

_MARK = "__inherit_ancestor_doc__"


def copy_ancestor_docstring(fn):
    """Decorator: take the docstring of the same-named method of the nearest documented ancestor class."""
    if fn.__doc__ is not None:
        raise RuntimeError("Function already has docstring")
    setattr(fn, _MARK, True)
    return fn


class InheritableDocstrings(ABCMeta):
    """Metaclass that resolves `copy_ancestor_docstring` marks when the class is created."""

    def __new__(mcls, name, bases, namespace, **kwargs):
        cls = super().__new__(mcls, name, bases, namespace, **kwargs)
        for attr, value in namespace.items():
            fn = getattr(value, "__func__", value)  # plain functions, classmethods and staticmethods
            if not getattr(fn, _MARK, False):
                continue
            for ancestor in cls.__mro__[1:]:
                doc = getattr(getattr(ancestor, attr, None), "__doc__", None)
                if doc is not None:
                    fn.__doc__ = doc
                    break
            else:
                raise RuntimeError(f"Can't inherit docstring for {attr}: method does not exist in superclass")
        return cls
