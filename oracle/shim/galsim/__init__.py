"""empty stand-in: only the module-level import in the reference needs to succeed"""
