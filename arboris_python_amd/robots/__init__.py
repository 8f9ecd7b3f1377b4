"""Model builders (inputs of the step): simplearm, snake, human36, single shapes."""
