"""Import-path shims so reference scripts (`from utils.X import Y`) find the MI355X implementations."""
