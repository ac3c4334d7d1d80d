#!/bin/bash
tools/round_profile.sh r04
