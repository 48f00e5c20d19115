#!/bin/sh
# regenerate the build system where autotools exist (they do not in the build image)
autoreconf --install
