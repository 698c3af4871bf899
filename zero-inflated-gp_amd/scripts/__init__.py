"""Look-alikes of the reference's experiment scripts (scripts/*.py) on the MI355X engine."""
